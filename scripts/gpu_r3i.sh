#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3i
timeout -k 10 600 python -m pytest tests -q -m gpu > gpurun_out/r3i/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r3i/pytest.log
timeout -k 10 900 bash scripts/profile_configs.sh r03 > gpurun_out/r3i/profile_configs.log 2>&1; tail -6 gpurun_out/r3i/profile_configs.log
timeout -k 10 600 bash scripts/pmc_extra.sh r03 > gpurun_out/r3i/pmc_extra.log 2>&1; tail -2 gpurun_out/r3i/pmc_extra.log | cut -c1-600
