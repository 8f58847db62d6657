#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3n; mkdir -p $O
timeout -k 10 500 python scripts/ab_bench.py --steps 30 --cycles 3 --out $O/ab.json trim=build/ab/libocc_trim.so fa=build/ab/libocc_fa.so ilp=build/ab/libocc_ilp.so mreg=build/ab/libocc_mreg.so > $O/ab.txt 2>&1; tail -5 $O/ab.txt
