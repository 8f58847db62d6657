#!/bin/bash
# A/B of raster-kernel builds on ONE box, interleaved rounds:  bash scripts/dbg/ab.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pool-models 64 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$name', 'raster %.3f ms' % j['roofline']['avg_launch_ms'], 'step %.3f ms' % j['ms_per_step'], '%.0f steps/s' % j['value'])"
}
for r in 1 2 3; do
  run old OCC_RASTER=1
  run newA X=1
  run newC OCC_HIP_LIB=$PWD/build/dbg2/libocc_vc.so
done
