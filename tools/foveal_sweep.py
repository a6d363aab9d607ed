#!/usr/bin/env python3
"""Launch-hint sweep of the foveal step kernels at 1M envs (fresh action row per step): one JSON line per variant with the
median of three interleaved passes per hint.   python tools/foveal_sweep.py v4 v5 [--auto-reset]"""
import importlib
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

lm = importlib.import_module("gym-lmaze_amd")
AR = "--auto-reset" in sys.argv
HINTS = [0x20, 0x25, 0x30, 0x34, 0x35, 0x36, 0x40, 0x43, 0x45, 0x120, 0x220, 0x130, 0x135, 0x230, 0x140, 0x145, 0x50]
N, rows = 1 << 20, 80
warmed = False
for variant in [a for a in sys.argv[1:] if not a.startswith("--")] or ["v4", "v5"]:
    env = lm.LmazeFovealVecEnv(N, variant=variant, seed=1)
    hi = 4 if variant in ("v1", "v5") else 25
    acts = torch.randint(0, hi, (rows, N), dtype=torch.int32, device="cuda")
    goals = torch.randint(0, 25, (rows, N), dtype=torch.int32, device="cuda")
    if variant == "v1":
        env.set_foveal_goal(torch.randint(0, 5, (N, 2), dtype=torch.int32, device="cuda"))
    if variant == "v5":
        env.foveal_done.fill_(True)
    k = [0]

    def run(n):
        for _ in range(n):
            r = k[0] % rows
            if variant == "v5":
                env.hier_step_raw(acts[r].data_ptr(), goals[r].data_ptr())
            else:
                env.step_raw(acts[r].data_ptr(), auto_reset=AR)
            k[0] += 1

    run(60 if warmed else 200)
    warmed = True
    res = {h: [] for h in [0] + HINTS}
    for _ in range(3):
        for h in res:
            env.params.launch_hint = h
            run(4)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(20)
            e1.record()
            torch.cuda.synchronize()
            res[h].append(e0.elapsed_time(e1) / 20 * 1e3)
    med = {("0x%x" % h): round(statistics.median(v), 1) for h, v in res.items()}
    env.params.launch_hint = 0
    print(json.dumps({"variant": variant, "envs": N, "auto_reset": AR, "default": lm._abi.describe_foveal_step(env.params, N, AR or variant == "v5"),
                      "best": sorted(med.items(), key=lambda kv: kv[1])[:5], "us": med}), flush=True)
    del env, acts, goals
    torch.cuda.empty_cache()
