"""Condenses the rocprofv3 CSVs of scripts/profile_round.sh into the small files committed under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

out, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)

stats = sorted(glob.glob(os.path.join(out, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)  # newest run
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(prof, f"{rnd}_kernel_stats.csv"), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:12]:
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def counters(sub):
    f = sorted(glob.glob(os.path.join(out, sub, "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)  # newest run
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f[0])):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


SHORT = "occ_raster2_kernel"
KEY = SHORT + "<true, true, true>"
summary = {"kernel": KEY, "kernel_short": SHORT, "note": "per-launch means over the full-batch step launches; FETCH_SIZE/WRITE_SIZE in KiB as "
           "rocprofv3 reports them; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 x2 read "
           "correction of MI355X_MICROARCH.md (HBM section), calibrated there for wide coalesced streams only"}
vals = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for k, v in counters(sub).items():
        if KEY in k:
            for c, xs in v.items():
                vals[c] = sum(xs) / len(xs)
                vals[c + "_launches"] = len(xs)
summary["counters"] = vals
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    summary["hbm_bytes_per_launch"] = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
try:
    bj = json.loads([l for l in open(os.path.join(out, "bench.json")) if l.startswith("{")][-1])
    summary["workload"] = "shapenet5k"
    summary["envs"] = bj["config"]["envs_per_gpu"]
    summary["img"] = bj["config"]["img"]
    summary["pool_models"] = bj["config"].get("pool_models", 64)
    summary["bench"] = {k: bj[k] for k in ("value", "ms_per_step", "roofline", "cpu_baseline") if k in bj}
    json.dump(bj, open(os.path.join(prof, f"{rnd}_bench.json"), "w"), indent=1)
except Exception as e:  # noqa: BLE001
    summary["bench_error"] = str(e)
json.dump(summary, open(os.path.join(prof, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(summary)[:1500])
