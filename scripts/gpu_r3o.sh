#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_rasterize_op.py -x -q -m gpu -s > $O/pytest.log 2>&1; echo "pytest rc $?"; grep "tiled" $O/pytest.log; tail -4 $O/pytest.log
