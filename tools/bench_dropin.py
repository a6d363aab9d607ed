#!/usr/bin/env python3
"""Latency of the single-env drop-in classes (BASELINE configs[0] / SURVEY C1: `make('lmaze-v0')`, N = 1):
steps/s of the reference-typed `step()` -- kernel launch + xE render + device->host copy of the
(C, G*E, G*E) float32 observation -- beside the reference interpreter's own rate recorded in BASELINE.md.

    python tools/bench_dropin.py [--steps 2000]

One JSON line per registered id.  This is the plumbing case, not the throughput metric (bench.py)."""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# reference interpreter, this build container, 1 core (BASELINE.md section 2 / SURVEY section 6)
REFERENCE_STEPS_PER_S = {"lmaze-v0": 74.0, "lmaze-v3": 120.0, "lmaze-v1": 370.0, "lmaze-v2": 345.0, "lmaze-v4": 263.0,
                         "lmaze-v5": 160.0, "lmaze-v6": 160.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--ids", default="lmaze-v0,lmaze-v3,lmaze-v1,lmaze-v2,lmaze-v4,lmaze-v5,lmaze-v6")
    args = ap.parse_args()
    import contextlib
    import io
    import torch
    import gym_lmaze
    random.seed(0)
    np.random.seed(0)
    for vid in args.ids.split(","):
        with contextlib.redirect_stdout(io.StringIO()):
            env = gym_lmaze.make(vid)
            env.reset()
        n_act = 25 if vid in ("lmaze-v2", "lmaze-v4") else 4
        acts = np.random.RandomState(1).randint(0, n_act, args.steps + 50)
        two_level = vid in ("lmaze-v5", "lmaze-v6")

        def one(t):
            a = int(acts[t])
            try:
                with contextlib.redirect_stdout(io.StringIO()) if vid == "lmaze-v4" else contextlib.nullcontext():
                    out = env.step(str(a) if vid == "lmaze-v3" else a)
            except IndexError:      # v5/v6: the reference's own local-view IndexError (lmaze_env_v5.py:364-365)
                return env.reset()
            done = out[4] if (two_level or vid == "lmaze-v1") else out[2]
            if two_level and out[5] and not done:              # localDone: the planner picks the next subgoal
                env.plannerStep(int(acts[t]) % 4)
            if done:
                env.reset()
            return out[0]

        for t in range(50):
            o = one(t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(50, 50 + args.steps):
            o = one(t)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rec = {"id": vid, "steps": args.steps, "us_per_step": dt / args.steps * 1e6, "steps_per_s": args.steps / dt,
               "obs_shape": list(np.asarray(o).shape), "includes": "launch + xE render + D2H obs + D2H scalars (+ reset on done)"}
        ref = REFERENCE_STEPS_PER_S.get(vid)
        if ref:
            rec["reference_steps_per_s"] = ref
            rec["x_reference"] = rec["steps_per_s"] / ref
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
