#!/bin/bash
# A/B on the mixed pool (1 280 - 20 480-face meshes, 1 024 envs): previous build vs current
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
run() {
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload mixed --envs 1024 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$name', 'raster %.3f ms' % j['roofline']['avg_launch_ms'], 'step %.3f ms' % j['ms_per_step'], '%.0f steps/s' % j['value'])"
}
for r in 1 2; do
  run before OCC_HIP_LIB=$PWD/build/dbg2/libocc_prev.so
  run after X=1
done
