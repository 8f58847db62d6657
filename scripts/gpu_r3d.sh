#!/bin/bash
# round 3, call D: tests of the split tiles / prologue changes, single-env latency, phase times of the raster kernel
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
timeout -k 10 300 python scripts/single_env_latency.py $O/single_env.json > $O/single_env.txt 2>&1; cat $O/single_env.txt | cut -c1-250
OCC_HIP_LIB=$PWD/build/ab/libocc_time.so timeout -k 10 200 python scripts/dbg/phase_time.py 1024 > $O/phase.txt 2>&1; cat $O/phase.txt | cut -c1-900
timeout -k 10 300 python scripts/ab_bench.py --steps 30 --cycles 2 --out $O/ab.json opt=build/ab/libocc_opt.so head=build/ab/libocc_head.so > $O/ab.txt 2>&1; tail -3 $O/ab.txt
