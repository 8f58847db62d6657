#!/bin/bash
# round 3, call C: A/B of the instruction-count work, full GPU tests on the new default build, 320-case parity sweep
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3c; mkdir -p $O
timeout -k 10 500 python scripts/ab_bench.py --steps 30 --cycles 2 --out $O/ab.json \
  new=build/ab/libocc_new.so noslp=build/ab/libocc_noslp.so opt=build/ab/libocc_opt.so opt_sw6=build/ab/libocc_opt_sw6.so \
  head=occlusionenv_amd/libocc_hip.so > $O/ab.txt 2>&1
tail -7 $O/ab.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
timeout -k 10 1500 python scripts/parity_sweep.py 320 1000 > $O/sweep.log 2>&1; echo "sweep rc $?" | tee -a $O/sweep.log
tail -3 $O/sweep.log
