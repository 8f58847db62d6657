#!/bin/bash
# Quick PMC passes for the raster kernel (GPU box):  bash scripts/dbg/pmc_quick.sh <tag> [bench args]
set -u
TAG=${1:-q}; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --pool-models 64 $*"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -- python "$ROOTD/bench.py" $ARGS > "$OUT/p$i.log" 2>&1)
done
python - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "p*", "*", "*counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "occ_raster" in r["Kernel_Name"] and "true, true, true" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, {c: "%.4g" % (sum(x) / len(x)) for c, x in v.items()})
PY
