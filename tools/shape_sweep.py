#!/usr/bin/env python3
"""Launch-policy sweep of the shared-layout step kernel for one or more shapes: envs per workgroup (launch_hint bits
10-11) x workgroups per CU x chunks per workgroup, plain and with the fused reset, a fresh action row per step from a ring
larger than the Infinity Cache.  The median of three interleaved passes counts.  One JSON line per (shape, mode):
    python tools/shape_sweep.py 11:v0:1048576 14:v0:1048576 18:v3:524288 32:v0:131072        (G:variant:envs)
This is the sweep behind the per-shape defaults of lmaze_step.hip launch_one / launch_shared."""
import importlib
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

lm = importlib.import_module("gym-lmaze_amd")
H = lm.LmazeVecEnv.launch_hint_of
POLICIES = ((8, 1), (6, 1), (5, 1), (4, 1), (3, 1), (2, 1), (8, 2), (5, 2), (4, 2), (3, 2), (2, 2), (4, 3))


def timed(env, acts, auto_reset, hint, steps=24):
    env.params.launch_hint = hint
    for i in range(6):
        env.step(acts[i % len(acts)], auto_reset=auto_reset)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        env.step(acts[i % len(acts)], auto_reset=auto_reset)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def main():
    shapes = sys.argv[1:] or ["11:v0:1048576"]
    warm = None
    for spec in shapes:
        G, variant, N = spec.split(":")
        G, N = int(G), int(N)
        rows = max(12, (320 << 20) // (4 * N))
        acts = torch.randint(0, 4, (rows, N), dtype=torch.int32, device="cuda")
        env = lm.LmazeVecEnv(N, variant=variant, layout=lm.layouts.open_room(G))
        if warm is None:                                   # cold clocks: the first ~100 launches of a process run slow
            for i in range(150):
                env.step(acts[i % rows])
            warm = True
        for auto_reset in (False, True):
            cands = [("default", 0)] + [("s%d %dx%d" % (sel, c, m), H(c, m, sel)) for sel in (1, 2, 3) for c, m in POLICIES]
            res = {k: [] for k, _ in cands}
            for _ in range(3):
                for name, h in cands:
                    res[name].append(timed(env, acts, auto_reset, h))
            med = {k: round(statistics.median(v), 1) for k, v in res.items()}
            B = 37 + 4 * G * G + (8 if variant == "v3" else 0)
            print(json.dumps({"G": G, "variant": variant, "envs": N, "auto_reset": auto_reset, "bytes_per_env_step": B,
                              "default_us": med["default"], "default_frac_of_8TBs": round(N * B / med["default"] / 8e6, 3),
                              "best": sorted(med.items(), key=lambda kv: kv[1])[:6], "us": med}), flush=True)
        del env, acts
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
