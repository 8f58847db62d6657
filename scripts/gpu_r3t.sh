#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3t; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_env_api.py -x -q -m gpu -k "ppo or config5 or graphed" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
OCC_BENCH_TRACE=1 timeout -k 10 300 python bench.py --workload ppo_rollout --steps 100 --warmup 5 --no-cpu-baseline > $O/ppo.json 2> $O/ppo.err; grep "trace" $O/ppo.err | cut -c1-700; python -c "
import json;j=json.loads([l for l in open('$O/ppo.json') if l.startswith('{')][-1]);print('ppo', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['ppo'])"
