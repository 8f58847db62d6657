"""profiles/<round>_<config>.json from the rocprofv3 output of scripts/profile_configs.sh (gpurun_out/<round>_cfg):
   python scripts/summarise_configs.py gpurun_out/r02_cfg r02
Runs on the GPU box at the end of profile_configs.sh and again locally (profiles/ written on the box does not travel back)."""
import csv, glob, json, os, sys

out, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for logf in sorted(glob.glob(os.path.join(out, "*.log"))):
    tag = os.path.basename(logf)[:-4]
    lines = [l for l in open(logf) if l.startswith("{")]
    if not lines:
        continue
    bench = json.loads(lines[-1])
    if tag == "n2_gloo":
        json.dump(bench, open(os.path.join(root, "profiles", f"{rnd}_n2_gloo_selflaunch.json"), "w"), indent=1)
        print(tag, "%.0f steps/s, %.2f ms/step" % (bench["value"], bench["ms_per_step"]))
        continue
    f = sorted(glob.glob(os.path.join(out, tag, "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)  # newest run
    rows = list(csv.DictReader(open(f[0])))[:8] if f else []
    rec = {"config": tag, "bench": bench, "kernels": [{k: r[k] for k in ("Name", "Calls", "AverageNs", "Percentage")} for r in rows]}
    unprof = os.path.join(out, tag + "_unprofiled.json")  # the same workload without the profiler (it taxes small launches)
    if os.path.exists(unprof):
        ul = [l for l in open(unprof) if l.startswith("{")]
        if ul:
            rec["bench_unprofiled"] = json.loads(ul[-1])
            rec["note"] = "bench = the run under rocprofv3 --kernel-trace (slower: the profiler taxes its small launches); bench_unprofiled = the same command without the profiler"
    json.dump(rec, open(os.path.join(root, "profiles", f"{rnd}_{tag}.json"), "w"), indent=1)
    print(tag, "%.0f steps/s, %.2f ms/step" % (bench["value"], bench["ms_per_step"]))
