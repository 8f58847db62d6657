#!/bin/bash
# Reproduces the committed profiles/ summaries for one round on a 1-GPU MI355X box:
#   bash scripts/profile_round.sh r01
# kernel-trace/stats and each PMC counter are collected in SEPARATE runs (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
R=${1:-r01}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/$R
mkdir -p "$OUT"
ARGS="--steps 10 --warmup 2 --no-cpu-baseline"
(cd /tmp && export TMPDIR=/tmp)
timeout -k 10 500 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python bench.py $ARGS > "$OUT/trace.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python bench.py $ARGS > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python bench.py $ARGS > "$OUT/pmc_write.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -- python bench.py $ARGS > "$OUT/pmc_sq.log" 2>&1
python scripts/summarise_profile.py "$OUT" "$R"
tail -1 "$OUT/bench.json" | cut -c1-600
