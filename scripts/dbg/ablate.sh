#!/bin/bash
# Ablation timings of the raster kernel (GPU box): bash scripts/dbg/ablate.sh  -> one "variant raster_ms step_ms" line each
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for v in "" NO_ATOM NO_SEL NO_EVAL; do
  if [ -n "$v" ]; then export OCC_HIP_LIB=$PWD/build/dbg2/libocc_$v.so; else unset OCC_HIP_LIB; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --pool-models 64 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('${v:-base}', 'raster %.3f ms' % j['roofline']['avg_launch_ms'], 'step %.3f ms' % j['ms_per_step'])"
done
