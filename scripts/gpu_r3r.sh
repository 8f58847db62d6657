#!/bin/bash
# round 3, final validation part 2 on HEAD: other configs, 320-case parity sweep
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3r; mkdir -p $O
timeout -k 10 500 bash scripts/profile_configs.sh r03 > $O/profile_configs.log 2>&1; tail -5 $O/profile_configs.log
timeout -k 10 400 python bench.py --workload ppo_rollout --steps 100 --warmup 5 --no-cpu-baseline > $O/ppo_unprofiled.json 2> $O/ppo.err; tail -c 500 $O/ppo_unprofiled.json
timeout -k 10 1000 python scripts/parity_sweep.py 320 1000 > $O/sweep.log 2>&1; echo "sweep rc $?" | tee -a $O/sweep.log; tail -3 $O/sweep.log | cut -c1-600
