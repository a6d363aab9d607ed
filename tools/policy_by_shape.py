#!/usr/bin/env python3
"""The shared-layout step kernel's launch policies across shapes (streaming regime), each shape on three separately
allocated envs (the placement of the observation buffer matters): v0 / v3, with and without the fused
reset.   python tools/policy_by_shape.py"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("gym-lmaze_amd")
POLICIES = ((3, 1), (3, 2), (4, 1), (4, 2), (5, 2), (8, 1), (2, 1), (2, 2))
K = 30
for variant, G, N in (("v0", 11, 1 << 20), ("v0", 8, 1 << 21), ("v0", 12, 1 << 20), ("v3", 18, 1 << 19), ("v0", 32, 1 << 17), ("v3", 11, 1 << 20)):
    R = max(2, (320 << 20) // (4 * N))
    ring = torch.randint(0, 4, (R, N), dtype=torch.int32, device="cuda")
    lay = pkg.layouts.to_codes(pkg.layouts.open_room(G, (G // 2, G // 2)))
    for auto_reset in (False, True):
        rows = []
        for inst in range(3):
            env = pkg.LmazeVecEnv(N, variant=variant, layout=lay, seed=1)
            pad = torch.empty((7 + 6 * inst) << 20, dtype=torch.uint8, device="cuda")
            t = 0
            for _ in range(150):
                env.step_raw(ring[t % R].data_ptr(), auto_reset=auto_reset); t += 1
            row = {}
            for pol in POLICIES:
                env.params.launch_hint = env.launch_hint_of(*pol)
                for _ in range(3):
                    env.step_raw(ring[t % R].data_ptr(), auto_reset=auto_reset); t += 1
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(K):
                    env.step_raw(ring[t % R].data_ptr(), auto_reset=auto_reset); t += 1
                e1.record()
                e1.synchronize()
                row["%dx%d" % pol] = round(e0.elapsed_time(e1) / K * 1e3, 1)
            rows.append(row)
            del env, pad
        print(json.dumps({"variant": variant, "G": G, "N": N, "auto_reset": auto_reset, "us": rows}), flush=True)
    del ring
    torch.cuda.empty_cache()
