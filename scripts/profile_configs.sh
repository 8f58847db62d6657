#!/bin/bash
# Kernel-time summaries of the other BASELINE configs on a 1-GPU box (config 2: teapot, 256 envs, 128x128, fwd+bwd;
# config-5 size: 2048 envs, 256x256, fwd+bwd; mixed pool) + the N>1 rehearsal (2 gloo ranks sharing the one GPU):
#   bash scripts/profile_configs.sh r03
set -u
R=${1:-r03}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/${R}_cfg
mkdir -p "$OUT" profiles
export TMPDIR=/tmp
PY=$(readlink -f "$(command -v python)")  # the interpreter itself after `--`, never a wrapper (scripts/pmc_kernels.sh)
run() {  # tag, bench args
  tag=$1; shift
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -- "$PY" "$ROOTD/bench.py" --no-cpu-baseline --fresh-steps 0 "$@" > "$OUT/$tag.log" 2>&1)
}
run config2_teapot256 --workload teapot --envs 256 --img 128 --steps 40 --warmup 5
run config5size_2048x256 --workload shapenet5k --envs 2048 --img 256 --steps 6 --warmup 2
run mixed1024 --workload mixed --envs 1024 --img 128 --steps 20 --warmup 3
# the reference's default image size, batched (environment.py:202, trainRL.py:75): 64 envs x 512x512
run config_512 --workload shapenet5k --envs 64 --img 512 --steps 20 --warmup 3
timeout -k 10 300 python bench.py --no-cpu-baseline --workload shapenet5k --envs 64 --img 512 --steps 60 --warmup 5 > "$OUT/config_512_unprofiled.json" 2> "$OUT/config_512_unprofiled.err"
# BASELINE config 5 at its per-rank size: PPO rollout (T = 50) + heads-only update, 256 envs, 256x256
run config5_rank --workload ppo_rollout --steps 100 --warmup 5
timeout -k 10 400 python bench.py --no-cpu-baseline --workload ppo_rollout --steps 200 --warmup 10 > "$OUT/config5_rank_unprofiled.json" 2> "$OUT/config5_rank_unprofiled.err"
# N>1 rehearsal from a COLD shell: bench.py starts torch.distributed.run itself (two gloo ranks sharing the single GPU:
# sharding, the side-stream record exchange, the gather)
timeout -k 10 400 python bench.py --gpus 2 --dist-backend gloo --envs 512 --steps 50 --warmup 5 > "$OUT/n2_gloo.log" 2>&1
python scripts/summarise_configs.py "$OUT" "$R"
