#!/usr/bin/env python3
"""Per-kernel means of the SQ counters of one rocprofv3 --pmc pass: python tools/pmc_sq.py <dir> [kernel-fragment]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
frag = sys.argv[2] if len(sys.argv) > 2 else "lmaze"
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:80]
        if frag not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k, c in acc.items():
    n = len(cnt[k])
    row = {name: v / n for name, v in c.items()}
    w = row.get("SQ_WAVES", 1.0)
    print(k, "dispatches", n)
    for name, v in sorted(row.items()):
        print("   %-22s %14.0f   per wave %10.1f" % (name, v, v / w))
