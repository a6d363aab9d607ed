#!/usr/bin/env python3
"""Which allocation decides the fast / slow state of a batch (placement_study2: it is a property of the env instance)?
Ten envs in one process; for each the (5,2) policy's time, the addresses of its state block and obs buffer; then env
A's state block driven with env B's obs buffer and vice versa.  python tools/placement_study3.py"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("gym-lmaze_amd")
N, G, K, R = 1 << 20, 11, 30, 80
lay = pkg.layouts.to_codes(pkg.layouts.open_room(G, (5, 5)))
ring = torch.randint(0, 4, (R, N), dtype=torch.int32, device="cuda")
pads, envs = [], []
for i in range(10):
    envs.append(pkg.LmazeVecEnv(N, variant="v0", layout=lay, seed=1))
    pads.append(torch.empty((3 + 5 * (i % 4)) << 20, dtype=torch.uint8, device="cuda"))
t = 0


def timed(env, pol):
    global t
    env.params.launch_hint = env.launch_hint_of(*pol)
    for _ in range(3):
        env.step_raw(ring[t % R].data_ptr()); t += 1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        env.step_raw(ring[t % R].data_ptr()); t += 1
    e1.record()
    e1.synchronize()
    return round(e0.elapsed_time(e1) / K * 1e3, 1)


for _ in range(300):
    envs[0].step_raw(ring[t % R].data_ptr()); t += 1
rows = []
for i, env in enumerate(envs):
    rows.append({"env": i, "state": hex(env._state.data_ptr()), "obs": hex(env.obs.data_ptr()), "5x2": timed(env, (5, 2)), "3x2": timed(env, (3, 2))})
fast = [r["env"] for r in rows if r["5x2"] < 90]
slow = [r["env"] for r in rows if r["5x2"] >= 90]
cross = {}
if fast and slow:
    a, b = envs[fast[0]], envs[slow[0]]
    pa, pb = a._p_obs, b._p_obs
    a._p_obs, b._p_obs = pb, pa          # swap the obs buffers only
    cross["fast state + slow env's obs"] = timed(a, (5, 2))
    cross["slow state + fast env's obs"] = timed(b, (5, 2))
    a._p_obs, b._p_obs = pa, pb
print(json.dumps({"rows": rows, "cross": cross}))
