#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3u; mkdir -p $O
timeout -k 10 500 bash scripts/profile_configs.sh r03 > $O/profile_configs.log 2>&1; tail -5 $O/profile_configs.log
timeout -k 10 300 python bench.py --workload ppo_rollout --steps 200 --warmup 10 --no-cpu-baseline > $O/ppo_unprofiled.json 2> $O/ppo.err; tail -c 400 $O/ppo_unprofiled.json
timeout -k 10 600 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
