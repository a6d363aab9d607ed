#!/usr/bin/env python3
"""Per-buffer duration and counters of one tools/placement_pmc.py pass under rocprofv3 (see there):
    python tools/placement_pmc_summary.py OUT_DIR [WARM] [RUNS]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
WARM = int(sys.argv[2]) if len(sys.argv) > 2 else 150
RUNS = int(sys.argv[3]) if len(sys.argv) > 3 else 14
trace = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
cnt = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
dur = {}
for f in trace:
    for r in csv.DictReader(open(f)):
        if "step_shared_kernel" in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
vals = defaultdict(dict)
for f in cnt:
    for r in csv.DictReader(open(f)):
        if "step_shared_kernel" in r["Kernel_Name"]:
            vals[int(r["Dispatch_Id"])][r["Counter_Name"]] = vals[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(set(dur) | set(vals))[WARM:]
out = []
for b in range(len(ids) // RUNS):
    grp = ids[b * RUNS + 2:(b + 1) * RUNS]
    row = {"buffer": b}
    if dur:
        row["us"] = round(sum(dur[i] for i in grp if i in dur) / max(1, sum(1 for i in grp if i in dur)), 2)
    names = sorted({n for i in grp for n in vals.get(i, {})})
    for n in names:
        row[n] = round(sum(vals[i].get(n, 0.0) for i in grp) / len(grp), 1)
    out.append(row)
out.sort(key=lambda r: r.get("us", 0))
for r in out:
    print(json.dumps(r))
