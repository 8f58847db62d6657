#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3j; mkdir -p $O
timeout -k 10 400 python scripts/ab_bench.py --steps 30 --cycles 3 --out $O/ab.json prev=build/ab/libocc_prev.so hst=build/ab/libocc_hst.so > $O/ab.txt 2>&1; tail -3 $O/ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
