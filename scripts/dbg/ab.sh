#!/bin/bash
# one-box interleaved A/B of configurations of the raster path (env toggles or OCC_HIP_LIB builds)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
run() {
  name=$1; shift
  env "$@" timeout -k 10 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --pool-models 64 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$name', 'raster %.3f ms' % j['roofline']['avg_launch_ms'], 'step %.3f ms' % j['ms_per_step'], '%.0f steps/s' % j['value'])"
}
for r in 1 2 3; do
  run before OCC_HIP_LIB=$PWD/build/dbg2/libocc_prev.so
  run after X=1
done
