#!/bin/bash
# Reproduces the committed profiles/ summaries for one round on a 1-GPU MI355X box:
#   bash scripts/profile_round.sh r02
# kernel-trace/stats and each PMC counter are collected in SEPARATE runs (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
R=${1:-r03}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
# rocprofv3 (with --pmc) initialises the GPU in the process it starts: what follows `--` must be the interpreter itself - an
# ELF binary - not a pyenv / conda shim or wrapper script that would exec again (forbidden on this pool once the GPU is open)
PY=$(readlink -f "$(command -v python)")
if [ "$(head -c 4 "$PY" | tail -c 3)" != "ELF" ]; then echo "python resolves to $PY, which is not an ELF binary: refusing to profile through a wrapper" >&2; exit 3; fi
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --fresh-steps 0"          # PMC passes: counters are per launch, every kernel is replayed
TRACE_ARGS="--steps 40 --warmup 4 --no-cpu-baseline --fresh-steps 0"    # --stats averages over the warm-up launches too: dilute them
timeout -k 10 500 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
prof() {  # subdir, rocprofv3 options..., then bench args after --
  d=$1; shift
  (cd /tmp && timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$d" -- "$PY" "$ROOTD/bench.py" $ARGS > "$OUT/$d.log" 2>&1)
}
SAVED=$ARGS; ARGS=$TRACE_ARGS
prof trace --kernel-trace --stats
ARGS=$SAVED
prof pmc_fetch --kernel-trace --pmc FETCH_SIZE
prof pmc_write --kernel-trace --pmc WRITE_SIZE
prof pmc_sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS
python scripts/summarise_profile.py "$OUT" "$R"
tail -1 "$OUT/bench.json" | cut -c1-600
