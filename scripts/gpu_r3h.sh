#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3h
timeout -k 10 1100 bash scripts/profile_round.sh r03 > gpurun_out/r3h/profile_round.log 2>&1; tail -3 gpurun_out/r3h/profile_round.log | cut -c1-700
