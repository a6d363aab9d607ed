#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/prof_rNN/) into the small files kept under
profiles/rNN/: per-kernel stats, the timed-region check, PMC traffic per launch (and profiles/traffic.json entries),
SQ counters.   python tools/profile_summarise.py <dir> <round>"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, rnd = sys.argv[1], int(sys.argv[2])
dst = os.path.join(src, "keep")
os.makedirs(dst, exist_ok=True)

KERNEL = {"c3": "step_shared_kernel<11, 0, true, ", "c2": "step_shared_wave8_kernel<0, true, ",
          "c5": "step_perenv_wave_kernel<32, 0, true, true>", "v1": "foveal_kernel<1, 0, ",
          "v2": "foveal_kernel<2, 0, ", "v4": "foveal_kernel<4, 0, ", "v5": "foveal_kernel<5, 0, "}   # envs per workgroup: tuned
KEY = {"c3": "g11_shared", "c5": "g32_perenv", "c2": "g8_shared"}


def line(path):
    try:
        return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        print("no bench line in", path, e)
        return None


for f in glob.glob(os.path.join(src, "bench_*.json")):
    d = line(f)
    if d:
        json.dump(d, open(os.path.join(dst, os.path.basename(f)), "w"), indent=1)

# --stats: per-kernel summary + the average launch duration of the workload's kernel against the bench's own events
for w in KERNEL:
    files = glob.glob(os.path.join(src, "stats_" + w, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        continue
    rows = list(csv.DictReader(open(files[0])))
    with open(os.path.join(dst, "bench_%s_kernel_stats.csv" % w), "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            wr.writerow([r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                         r["MaxNs"], r["StdDev"]])
    hit = [r for r in rows if KERNEL[w] in r["Name"]]
    b = line(os.path.join(src, "bench_%s_under_rocprofv3.json" % w))
    if hit and b:
        avg_us = float(hit[0]["AverageNs"]) / 1e3
        N, B = b["config"]["envs_per_gpu"], b["roofline"]["bytes_per_env_step"]
        rec = {"workload": w, "kernel": hit[0]["Name"][:100], "rocprofv3_stats_calls": int(hit[0]["Calls"]),
               "rocprofv3_stats_avg_us": avg_us, "bench_events_avg_us_same_process": b["roofline"]["kernel_ms_avg"] * 1e3,
               "frac_of_8TBs_from_stats_avg": N * B / (avg_us * 1e-6) / 8e12, "frac_reported_by_bench_same_process": b["roofline"]["frac"],
               "launch_hint": b["config"]["launch_hint"],
               "note": "every profiled launch of this kernel runs the policy of the timed region (no autotune in the process); "
                       "the device is warmed with a different kernel (400 fill-probe launches), so the --stats average covers --warmup + --steps warm launches of this kernel, the bench events the --steps timed ones"}
        # timed region from the trace: the last `steps` launches of the kernel
        tr = glob.glob(os.path.join(src, "stats_" + w, "**", "*kernel_trace.csv"), recursive=True)
        if tr:
            d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(tr[0])) if KERNEL[w] in r["Kernel_Name"]]
            d.sort()
            last = d[-b["steps"]:]
            rec["rocprofv3_trace_last_%d_launches_avg_us" % b["steps"]] = sum(e - s for s, e in last) / len(last) / 1e3
        json.dump(rec, open(os.path.join(dst, "bench_%s_timed_region.json" % w), "w"), indent=1)
        print(json.dumps(rec))

# PMC traffic
for w in KERNEL:
    dw, df = os.path.join(src, "pmc_%s_WRITE_SIZE" % w), os.path.join(src, "pmc_%s_FETCH_SIZE" % w)
    b = line(os.path.join(src, "pmc_%s_WRITE_SIZE.json" % w))
    if not (os.path.isdir(dw) and os.path.isdir(df) and b):
        continue
    alg = int(round(b["config"]["envs_per_gpu"] * b["roofline"]["bytes_per_env_step"]))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), "--write", dw, "--fetch", df, "--kernel",
                          KERNEL[w], "--key", KEY.get(w, w), "--algorithmic", str(alg), "--round", str(rnd), "--out-prefix",
                          os.path.join(dst, "bench_" + w)], capture_output=True, text=True)
    print(w, out.stdout[-400:], out.stderr[-400:])
# the box's profiles/traffic.json does not travel back (only gpurun_out/ is merged): keep a copy beside the summaries
tpath = os.path.join(ROOT, "profiles", "traffic.json")
if os.path.exists(tpath):
    json.dump(json.load(open(tpath)), open(os.path.join(dst, "traffic.json"), "w"), indent=1)

# SQ counters
sq = {}
for w in KERNEL:
    files = glob.glob(os.path.join(src, "sq_" + w, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    acc, ids = defaultdict(float), set()
    for r in csv.DictReader(open(files[0])):
        if KERNEL[w] in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            ids.add(r["Dispatch_Id"])
    n = max(1, len(ids))
    waves = acc.get("SQ_WAVES", 0) / n
    sq[w] = {"kernel": KERNEL[w], "dispatches": len(ids), "per_launch": {k: v / n for k, v in sorted(acc.items())},
             "per_wave": {k: v / n / waves for k, v in sorted(acc.items())} if waves else None}
if sq:
    json.dump(sq, open(os.path.join(dst, "sq_counters.json"), "w"), indent=1)
    print(json.dumps({w: v["per_wave"] for w, v in sq.items()}))
