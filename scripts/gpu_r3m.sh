#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3m; mkdir -p $O
timeout -k 10 400 python scripts/ab_bench.py --steps 30 --cycles 3 --out $O/ab.json cur=build/ab/libocc_cur.so trim=build/ab/libocc_trim.so > $O/ab.txt 2>&1; tail -3 $O/ab.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
