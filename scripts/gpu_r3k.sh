#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3k; mkdir -p $O
timeout -k 10 400 python scripts/ab_bench.py --steps 30 --cycles 3 --out $O/ab.json sorted=build/ab/libocc_sorted.so mix=build/ab/libocc_mix.so > $O/ab.txt 2>&1; tail -3 $O/ab.txt
timeout -k 10 400 python scripts/ab_bench.py --steps 20 --cycles 2 --workload mixed --out $O/ab_mixed.json sorted=build/ab/libocc_sorted.so mix=build/ab/libocc_mix.so > $O/ab_mixed.txt 2>&1; tail -3 $O/ab_mixed.txt
timeout -k 10 400 python scripts/ab_bench.py --steps 20 --cycles 2 --envs 256 --img 256 --out $O/ab_256.json sorted=build/ab/libocc_sorted.so mix=build/ab/libocc_mix.so > $O/ab_256.txt 2>&1; tail -3 $O/ab_256.txt
