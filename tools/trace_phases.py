#!/usr/bin/env python3
"""Split a rocprofv3 kernel trace of `bench.py` into the bench's phases and average the step kernel per phase.

    python tools/trace_phases.py <..._kernel_trace.csv> [--steps 300] [--kernel 'lmaze::step_']

The --stats summary averages EVERY launch of the kernel in the process: autotune's warm-up launches (cold
clocks), its candidates under other launch policies, the bench's own warm-up, and the timed region.  Only the
last `steps` launches are the timed region bench.py reports; this prints each part so that the two can be
compared like for like."""
import argparse
import csv
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--kernel", default="lmaze::step_")
    args = ap.parse_args()
    rows = [r for r in csv.DictReader(open(args.trace)) if args.kernel in r["Kernel_Name"]]
    # the step launches proper (DO_STEP = true is the third template argument; observe launches are not steps)
    steps = [r for r in rows if ", true," in r["Kernel_Name"].split("(")[0]]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in steps]
    timed, before = d[-args.steps:], d[:-args.steps]
    out = {"kernel": steps[-1]["Kernel_Name"].split("(")[0], "launches_total": len(d),
           "avg_ns_all_launches": sum(d) / len(d),
           "timed_region": {"launches": len(timed), "avg_ns": sum(timed) / len(timed), "min_ns": min(timed),
                            "max_ns": max(timed)},
           "before_timed_region": {"launches": len(before), "avg_ns": (sum(before) / len(before)) if before else None,
                                   "what": "autotune warm-up + candidates, bench warm-up"}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
