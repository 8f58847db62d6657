#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3l; mkdir -p $O
timeout -k 10 400 python scripts/ab_bench.py --steps 30 --cycles 3 --out $O/ab.json cur=build/ab/libocc_cur.so da=build/ab/libocc_da.so > $O/ab.txt 2>&1; tail -3 $O/ab.txt
OCC_HIP_LIB=$PWD/build/ab/libocc_da.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest(da) rc $?"; tail -3 $O/pytest.log
