#!/usr/bin/env python3
"""Second placement question: the per-env STATE block and the action ring.  Four envs (separate state + obs
allocations) and two action rings in one process, the same three launch policies on each.  python tools/placement_study2.py"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("gym-lmaze_amd")
N, G, K, R = 1 << 20, 11, 40, 80
lay = pkg.layouts.to_codes(pkg.layouts.open_room(G, (5, 5)))
rings = [torch.randint(0, 4, (R, N), dtype=torch.int32, device="cuda") for _ in range(2)]
pads, envs = [], []
for i in range(4):
    envs.append(pkg.LmazeVecEnv(N, variant="v0", layout=lay, seed=1))
    pads.append(torch.empty((3 + 5 * i) << 20, dtype=torch.uint8, device="cuda"))     # shift the next env's allocations
t = 0
out = {}
for _ in range(300):
    envs[0].step_raw(rings[0][t % R].data_ptr()); t += 1
for rnd in range(2):
    for ei, env in enumerate(envs):
        for ri, ring in enumerate(rings):
            for pol in ((3, 2), (5, 2), (8, 2)):
                env.params.launch_hint = env.launch_hint_of(*pol)
                for _ in range(3):
                    env.step_raw(ring[t % R].data_ptr()); t += 1
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(K):
                    env.step_raw(ring[t % R].data_ptr()); t += 1
                e1.record()
                e1.synchronize()
                out.setdefault("env%d ring%d %dx%d" % (ei, ri, pol[0], pol[1]), []).append(round(e0.elapsed_time(e1) / K * 1e3, 1))
print(json.dumps({"state_ptr_mod_2MiB": [hex(e._state.data_ptr() % (2 << 20)) for e in envs], "us": out}))
