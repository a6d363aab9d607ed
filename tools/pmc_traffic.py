#!/usr/bin/env python3
"""HBM bytes per launch of one kernel from two rocprofv3 PMC passes (WRITE_SIZE and FETCH_SIZE are collected
in SEPARATE runs: they do not fit the TCC counter slots together, and gpurun refuses --pmc combined with the
hip/hsa trace domains).

    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dirW> -- python3 bench.py --workload v2 --steps 20 --warmup 5 --no-cpu-baseline
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dirF> -- python3 bench.py --workload v2 --steps 20 --warmup 5 --no-cpu-baseline
    python tools/pmc_traffic.py --write <dirW> --fetch <dirF> --kernel 'foveal_kernel<2, 0' --key v2 --algorithmic <bytes>

Both counters are in KB (x1024).  FETCH_SIZE is doubled, as guides/MI355X_MICROARCH.md prescribes for gfx950 (it
tallies 128-byte read requests at 64 B; calibrated on this library's observe launch, profiles/traffic.json _note).
Writes per-kernel summaries next to the raw files and merges the result into profiles/traffic.json under --key."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel_mean(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under %s" % directory
    per_dispatch = defaultdict(float)
    name_of = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            key = (r.get("Process_Id", ""), r["Dispatch_Id"])
            per_dispatch[key] += float(r["Counter_Value"])          # one row per counter instance
            name_of[key] = r["Kernel_Name"]
    by_kernel = defaultdict(list)
    for key, v in per_dispatch.items():
        by_kernel[name_of[key]].append(v)
    return {k: (len(v), sum(v) / len(v)) for k, v in by_kernel.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", required=True)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--kernel", required=True, help="substring of the kernel name")
    ap.add_argument("--key", required=True, help="entry of profiles/traffic.json")
    ap.add_argument("--algorithmic", type=int, required=True, help="algorithmic bytes per launch")
    ap.add_argument("--round", type=int, default=1)
    ap.add_argument("--out-prefix", default=None, help="write <prefix>_write_size.csv / _fetch_size.csv summaries")
    args = ap.parse_args()
    w, f = per_kernel_mean(args.write, "WRITE_SIZE"), per_kernel_mean(args.fetch, "FETCH_SIZE")
    if args.out_prefix:
        for tag, d, cname in (("write", w, "WRITE_SIZE"), ("fetch", f, "FETCH_SIZE")):
            with open("%s_%s_size.csv" % (args.out_prefix, tag), "w", newline="") as fh:
                wr = csv.writer(fh)
                wr.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Mean_Counter_Value_KB"])
                for k, (n, m) in sorted(d.items(), key=lambda kv: -kv[1][1]):
                    wr.writerow([k[:100], cname, n, "%.3f" % m])
    kw = [k for k in w if args.kernel in k]
    kf = [k for k in f if args.kernel in k]
    assert len(kw) == 1 and len(kf) == 1, (kw, kf)
    wkb, fkb = w[kw[0]][1], f[kf[0]][1]
    hbm = int(round(wkb * 1024 + 2 * fkb * 1024))
    rec = {"round": args.round, "kernel": kw[0].split("(")[0][:100], "write_size_kb": wkb, "fetch_size_kb": fkb,
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": args.algorithmic,
           "dispatches": [w[kw[0]][0], f[kf[0]][0]],
           "files": "profiles/r%02d/%s_{write,fetch}_size.csv" % (args.round, os.path.basename(args.out_prefix) if args.out_prefix else args.key)}
    path = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(path)) if os.path.exists(path) else {}
    tj[args.key] = rec
    json.dump(tj, open(path, "w"), indent=1)
    print(json.dumps(rec, indent=1), "\nratio measured / algorithmic = %.4f" % (hbm / args.algorithmic))


if __name__ == "__main__":
    main()
