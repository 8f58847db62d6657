#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3e; mkdir -p $O
timeout -k 10 500 python scripts/ab_bench.py --steps 30 --cycles 2 --out $O/ab.json head=build/ab/libocc_head.so h2=build/ab/libocc_h2.so \
   fp2=build/ab/libocc_fp2.so fp3=build/ab/libocc_fp3.so fp4=build/ab/libocc_fp4.so > $O/ab.txt 2>&1; tail -6 $O/ab.txt
OCC_HIP_LIB=$PWD/build/ab/libocc_opt.so timeout -k 10 300 python scripts/single_env_latency.py $O/single_env_before.json > $O/single_env_before.txt 2>&1; cut -c1-200 $O/single_env_before.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
