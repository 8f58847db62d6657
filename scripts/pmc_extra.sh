#!/bin/bash
# Extra PMC passes for the raster kernel (instruction mix, LDS, L2): writes profiles/<round>_pmc_extra.json.
#   bash scripts/pmc_extra.sh r01        (on a 1-GPU MI355X box; counters in separate runs, kernel-trace only)
set -u
R=${1:-r03}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/${R}_extra
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OLDPWD/$OUT/p$i" -- python "$OLDPWD/bench.py" --steps 6 --warmup 2 --no-cpu-baseline > "$OLDPWD/$OUT/p$i.log" 2>&1)
done
python - "$OUT" "$R" <<'PY'
import collections, csv, glob, json, os, sys
out, rnd = sys.argv[1], sys.argv[2]
KEY = "occ_raster2_kernel<true, true, true>"
vals = {}
for f in glob.glob(os.path.join(out, "p*", "*", "*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if KEY in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        vals[k] = sum(v) / len(v)
json.dump({"kernel": KEY, "note": "per-launch means, default bench workload", "counters": vals},
          open(os.path.join("profiles", f"{rnd}_pmc_extra.json"), "w"), indent=1)
print(json.dumps(vals))
PY
