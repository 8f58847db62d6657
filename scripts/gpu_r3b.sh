#!/bin/bash
# round 3, call B: second A/B batch, full GPU tests, kernel trace of the bench for the per-step launch sequence
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python scripts/ab_bench.py --steps 30 --cycles 2 --out $O/ab.json \
  new=build/ab/libocc_new.so noslp=build/ab/libocc_noslp.so ls3=build/ab/libocc_ls3.so ls4_12=build/ab/libocc_ls4.so:12 \
  c2ls4_14=build/ab/libocc_c2ls4.so:14 c2ls4_16=build/ab/libocc_c2ls4.so:16 > $O/ab.txt 2>&1
tail -8 $O/ab.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
ROOTD=$PWD
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/$O/trace -- python $ROOTD/bench.py --steps 12 --warmup 4 --no-cpu-baseline > $ROOTD/$O/trace.log 2>&1)
python scripts/trace_gaps.py $O/trace $O/trace_gaps.json > $O/trace_gaps.txt 2>&1; head -60 $O/trace_gaps.txt
