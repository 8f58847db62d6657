#!/bin/bash
# PMC passes over the bench for EVERY kernel of the step (setup, raster, combine, ...): per-launch means per kernel.
#   bash scripts/pmc_kernels.sh <tag> [bench args]     -> gpurun_out/<tag>/pmc_kernels.json  (1-GPU MI355X box; each
# counter set in a run of its own with --kernel-trace only, as MI355X_MICROARCH.md prescribes)
set -u
TAG=${1:-pmc}; shift || true
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# rocprofv3 (with --pmc) initialises the GPU in the process it starts: what follows `--` must be the interpreter itself - an
# ELF binary - not a pyenv / conda shim or wrapper script that would exec again (forbidden on this pool once the GPU is open)
PY=$(readlink -f "$(command -v python)")
if [ "$(head -c 4 "$PY" | tail -c 3)" != "ELF" ]; then echo "python resolves to $PY, which is not an ELF binary: refusing to profile through a wrapper" >&2; exit 3; fi
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -- "$PY" "$ROOTD/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --fresh-steps 0 "$@" > "$OUT/p$i.log" 2>&1)
  echo "pmc pass $i rc $?"
done
python - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "occ::" not in name and "occ_" not in name:
            continue
        short = name.split("(")[0].replace("void ", "").replace("occ::", "")
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in agg.items():
    # full-batch launches only: keep the upper half by SQ_WAVES-independent proxy (largest values of the first counter)
    res[k] = {}
    for c, xs in cs.items():
        xs = sorted(xs)
        big = [x for x in xs if x >= 0.5 * xs[-1]] if xs[-1] > 0 else xs
        res[k][c] = sum(big) / len(big)
        res[k][c + "_launches"] = len(big)
json.dump(res, open(os.path.join(out, "pmc_kernels.json"), "w"), indent=1)
for k in sorted(res):
    v = res[k]
    if "SQ_WAVE_CYCLES" in v:
        print(k[:60], {c: ("%.3g" % v[c]) for c in sorted(v) if not c.endswith("_launches")})
PY
