#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3s; mkdir -p $O
OCC_BENCH_TRACE=1 timeout -k 10 300 python bench.py --workload ppo_rollout --steps 110 --warmup 5 --no-cpu-baseline > $O/ppo.json 2> $O/ppo.err; grep "trace" $O/ppo.err | cut -c1-1200; tail -c 300 $O/ppo.json
timeout -k 10 300 python bench.py --envs 256 --img 256 --steps 100 --warmup 5 --no-cpu-baseline > $O/plain256.json 2>/dev/null; python -c "
import json;j=json.loads([l for l in open('$O/plain256.json') if l.startswith('{')][-1]);print('plain 256x256x256 envs', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"
nproc; python -c "import os;print(len(os.sched_getaffinity(0)))"
