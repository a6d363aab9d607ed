#!/usr/bin/env python3
"""Sweep LmazeFovealParams.launch_hint (envs per workgroup x workgroups per CU) on the foveal step kernels at 1M envs,
fresh action row per step from a ring larger than the Infinity Cache (the bench's regime).  Interleaved rounds, the
minimum over rounds counts.  python tools/foveal_hint_study.py [variants...]  -> one JSON line per variant."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("gym-lmaze_amd")
N, K = 1 << 20, 60
variants = sys.argv[1:] or ["v1", "v2", "v4", "v5"]
hints = [0] + [(e << 4) | c for e in (2, 3, 4, 5) for c in (0, 2, 3, 4, 6)]
if os.environ.get('HINTS'):
    hints = [int(x, 16) for x in os.environ['HINTS'].split(',')]
else:
    hints += [(m << 8) | (e << 4) | c for m in (1, 2, 3) for e in (2, 3) for c in (0, 4, 6)]      # bits 8-9: chunks per workgroup - 1
for variant in variants:
    env = pkg.LmazeFovealVecEnv(N, variant=variant, seed=1)
    hi = 4 if variant in ("v1", "v5") else 25
    R = 80
    acts = torch.randint(0, hi, (R, N), dtype=torch.int32, device="cuda")
    goals = torch.randint(0, 25, (R, N), dtype=torch.int32, device="cuda")
    if variant == "v1":
        env.set_foveal_goal(torch.randint(0, 5, (N, 2), dtype=torch.int32, device="cuda"))
    if variant == "v5":
        env.foveal_done.fill_(True)
    ap = [acts[r].data_ptr() for r in range(R)]
    gp = [goals[r].data_ptr() for r in range(R)]
    t = 0

    def run(k):
        global t
        for _ in range(k):
            if variant == "v5":
                env.hier_step_raw(ap[t % R], gp[t % R])
            else:
                env.step_raw(ap[t % R])
            t += 1

    run(200)
    best = {}
    for rnd in range(3):
        for h in hints:
            env.params.launch_hint = h
            run(3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(K)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / K
            best[h] = min(ms, best.get(h, ms))
    print(json.dumps({"variant": variant, "envs": N, "ms_by_hint": {"0x%03x" % h: round(v, 4) for h, v in best.items()},
                      "best": "0x%03x" % min(best, key=best.get)}), flush=True)
    del env, acts, goals
    torch.cuda.empty_cache()
