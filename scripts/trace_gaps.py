#!/usr/bin/env python
"""Per-step launch sequence from a rocprofv3 --kernel-trace CSV: kernels between two consecutive full-batch raster
launches, with start offsets, durations and the idle gaps between them.

  python scripts/trace_gaps.py <dir with *_kernel_trace.csv> [out.json]
"""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = []
for r in csv.DictReader(open(files[-1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
KEY = "occ_raster2_kernel<true, true, true>"
big = [i for i, r in enumerate(rows) if KEY in r[2]]
durs = sorted(rows[i][1] - rows[i][0] for i in big)
full = [i for i in big if rows[i][1] - rows[i][0] > 0.5 * durs[-1]]  # full-batch launches only
steps = []
for a, b in zip(full[:-1], full[1:]):
    seq = rows[a:b]
    t0 = seq[0][0]
    busy = sum(e - s for s, e, _ in seq)
    span = rows[b][0] - t0
    steps.append(dict(span_us=span / 1e3, busy_us=busy / 1e3, idle_us=(span - busy) / 1e3, n_kernels=len(seq),
                      raster_us=(seq[0][1] - seq[0][0]) / 1e3))
mid = steps[len(steps) // 2:]
avg = {k: sum(s[k] for s in mid) / len(mid) for k in mid[0]}
print("steady-state step (mean over the last %d): %s" % (len(mid), json.dumps({k: round(v, 1) for k, v in avg.items()})))
a, b = full[-2], full[-1]
t0 = rows[a][0]
prev_end = t0
listing = []
for s, e, n in rows[a:b]:
    short = n.split("(")[0].replace("void ", "")[:70]
    listing.append(dict(at_us=round((s - t0) / 1e3, 1), dur_us=round((e - s) / 1e3, 1), gap_us=round((s - prev_end) / 1e3, 1), kernel=short))
    prev_end = max(prev_end, e)
for l in listing:
    print("%9.1f  +%7.1f  gap %6.1f  %s" % (l["at_us"], l["dur_us"], l["gap_us"], l["kernel"]))
if len(sys.argv) > 2:
    json.dump(dict(steady_state=avg, last_step=listing), open(sys.argv[2], "w"), indent=1)
