#!/bin/bash
# round 3, final validation on HEAD: full GPU tests, profile set (bench + PMC + configs + single env), 320-case parity sweep
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3q; mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout -k 10 600 bash scripts/profile_round.sh r03 > $O/profile_round.log 2>&1; tail -1 $O/profile_round.log | cut -c1-400
timeout -k 10 500 bash scripts/pmc_extra.sh r03 > $O/pmc_extra.log 2>&1; tail -1 $O/pmc_extra.log | cut -c1-300
timeout -k 10 300 python scripts/single_env_latency.py $O/single_env.json > $O/single_env.txt 2>&1; cut -c1-200 $O/single_env.txt
