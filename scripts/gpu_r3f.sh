#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3f; mkdir -p $O
timeout -k 10 500 python scripts/ab_bench.py --steps 30 --cycles 2 --out $O/ab.json h2=build/ab/libocc_h2.so fp4=build/ab/libocc_fp4.so \
   s2=build/ab/libocc_s2.so s2fp2=build/ab/libocc_s2fp2.so s2fp4=build/ab/libocc_s2fp4.so > $O/ab.txt 2>&1; tail -6 $O/ab.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
