#!/usr/bin/env python3
"""Does the placement of the observation buffer matter?  From process to process the same C3 launch policies time
differently -- (4,2)...(8,2) at 81-85 us in one process and 99-102 in the next, (3,2) at 83 in both -- which smells of
where the allocation landed (channel hashing of the 507-MB write stream against the 21 MB of inputs).  One process:
the step kernel on the same state, the obs pointer moved over offsets inside one large pool and over separately
allocated buffers.   python tools/placement_study.py"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("gym-lmaze_amd")
N, G, K = 1 << 20, 11, 40
env = pkg.LmazeVecEnv(N, variant="v0", layout=pkg.layouts.to_codes(pkg.layouts.open_room(G, (5, 5))), seed=1)
R = 80
acts = torch.randint(0, 4, (R, N), dtype=torch.int32, device="cuda")
ap = [acts[r].data_ptr() for r in range(R)]
nbytes = N * G * G * 4
pool = torch.empty(nbytes * 3 // 4 + (64 << 20), dtype=torch.int32, device="cuda")       # 3x the buffer + slack, in ints
others = [torch.empty(N * G * G, dtype=torch.int32, device="cuda") for _ in range(3)]
t = 0


def run(k):
    global t
    for _ in range(k):
        env.step_raw(ap[t % R])
        t += 1


run(300)
spots = [("env.obs", env.obs.data_ptr())]
for off in (0, 4096, 65536, 1 << 20, (1 << 21) + 65536, 16 << 20, (32 << 20) + 8192, nbytes + 4096, nbytes + (3 << 20)):
    spots.append(("pool+%d" % off, pool.data_ptr() + off))
spots += [("alloc%d" % i, b.data_ptr()) for i, b in enumerate(others)]
out = {}
for rnd in range(2):
    for name, ptr in spots:
        for pol in ((3, 2), (5, 2), (8, 2)):
            env.params.launch_hint = env.launch_hint_of(*pol)
            env._p_obs = ptr
            run(3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(K)
            e1.record()
            e1.synchronize()
            key = "%s %dx%d" % (name, pol[0], pol[1])
            out.setdefault(key, []).append(round(e0.elapsed_time(e1) / K * 1e3, 1))
print(json.dumps({"ptr_mod_2MiB": {n: hex(p % (2 << 20)) for n, p in spots}, "us": out}, indent=0))
