#!/usr/bin/env python3
"""How long must a launch-policy candidate run before its timing is the steady-state one?
Times every workgroups-per-CU candidate of the C3 step kernel over bursts of different length."""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("gym-lmaze_amd")

N, G = 1 << 20, 11
env = pkg.LmazeVecEnv(N, variant="v0", layout=pkg.layouts.to_codes(pkg.layouts.open_room(G, (5, 5))), online_autotune=False)
a = torch.randint(0, 4, (N,), dtype=torch.int32, device="cuda")


def burst(hint, steps):
    env.params.launch_hint = hint
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        env.step_raw(a.data_ptr())
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


for _ in range(200):
    env.step_raw(a.data_ptr())
torch.cuda.synchronize()
for rnd in range(3):
    for steps in (4, 8, 20, 50, 150, 400):
        row = {"round": rnd, "steps": steps}
        for hint in (8, 4, 3, 2):
            row[str(hint)] = round(burst(hint, steps), 2)
        print(json.dumps(row), flush=True)
