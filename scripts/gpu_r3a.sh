#!/bin/bash
# round 3, call A: full GPU test suite, A/B of the first kernel variants, the self-launching N = 2 gloo rehearsal, config 5 per rank
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout -k 10 600 python scripts/ab_bench.py --steps 30 --cycles 2 --out $O/ab.json \
  r02=build/ab/libocc_r02.so new=build/ab/libocc_new.so sync=build/ab/libocc_sync.so w4=build/ab/libocc_w4.so \
  c2=build/ab/libocc_c2.so c2w4_12=build/ab/libocc_c2w4.so:12 c2w4_14=build/ab/libocc_c2w4.so:14 c2w4_16=build/ab/libocc_c2w4.so:16 > $O/ab.txt 2>&1
tail -12 $O/ab.txt
timeout -k 10 400 python bench.py --gpus 2 --dist-backend gloo --envs 512 --steps 50 --warmup 5 > $O/n2_gloo.json 2> $O/n2_gloo.err; echo "n2 rc $?"
tail -c 600 $O/n2_gloo.json
timeout -k 10 400 python bench.py --workload ppo_rollout --steps 100 --warmup 5 --no-cpu-baseline > $O/ppo.json 2> $O/ppo.err; echo "ppo rc $?"
tail -c 900 $O/ppo.json
