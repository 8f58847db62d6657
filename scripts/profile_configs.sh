#!/bin/bash
# Kernel-time summaries of the other BASELINE configs on a 1-GPU box (config 2: teapot, 256 envs, 128x128, fwd+bwd;
# config-5 size: 2048 envs, 256x256, fwd+bwd; mixed pool) + the N>1 rehearsal (2 gloo ranks sharing the one GPU):
#   bash scripts/profile_configs.sh r02
set -u
R=${1:-r02}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/${R}_cfg
mkdir -p "$OUT" profiles
export TMPDIR=/tmp
run() {  # tag, bench args
  tag=$1; shift
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -- python "$ROOTD/bench.py" --no-cpu-baseline "$@" > "$OUT/$tag.log" 2>&1)
}
run config2_teapot256 --workload teapot --envs 256 --img 128 --steps 40 --warmup 5
run config5size_2048x256 --workload shapenet5k --envs 2048 --img 256 --steps 6 --warmup 2
run mixed1024 --workload mixed --envs 1024 --img 128 --steps 20 --warmup 3
# N>1 rehearsal: two ranks (gloo) sharing the single GPU: exercises sharding, the side-stream record exchange, the gather
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --dist-backend gloo --envs 512 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/n2_gloo.log" 2>&1
python scripts/summarise_configs.py "$OUT" "$R"
