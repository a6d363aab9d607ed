#!/usr/bin/env python3
"""placement_study3 for the foveal envs and the per-env-layout batch: several instances in one process, the uncapped
policy's time on each, then the obs buffers of the fastest and the slowest swapped.  python tools/placement_study4.py v2 v1"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("gym-lmaze_amd")
N, K, R = 1 << 20, 30, 80
for variant in (sys.argv[1:] or ["v2", "v1"]):
    hi = 4 if variant in ("v1", "v5") else 25
    ring = torch.randint(0, hi, (R, N), dtype=torch.int32, device="cuda")
    envs, pads = [], []
    for i in range(8 if variant != "v4" else 6):
        e = pkg.LmazeFovealVecEnv(N, variant=variant, seed=1)
        if variant == "v1":
            e.set_foveal_goal(torch.randint(0, 5, (N, 2), dtype=torch.int32, device="cuda"))
        envs.append(e)
        pads.append(torch.empty((3 + 5 * (i % 4)) << 20, dtype=torch.uint8, device="cuda"))
    t = 0

    def timed(env, hint):
        global t
        env.params.launch_hint = hint
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
    rows = [{"env": i, "0x30": timed(e, 0x30), "0x35": timed(e, 0x35), "0x34": timed(e, 0x34)} for i, e in enumerate(envs)]
    a = min(rows, key=lambda r: r["0x30"])["env"]
    b = max(rows, key=lambda r: r["0x30"])["env"]
    ea, eb = envs[a], envs[b]
    pa, pb = ea.bufs.obs, eb.bufs.obs
    ea.bufs.obs, eb.bufs.obs = pb, pa
    cross = {"fastest env %d with the obs of the slowest %d" % (a, b): timed(ea, 0x30), "slowest with the obs of the fastest": timed(eb, 0x30)}
    ea.bufs.obs, eb.bufs.obs = pa, pb
    if envs[0].visit is not None:        # v4: the visit maps are a second, larger stream (read + write)
        va, vb = ea.bufs.visit, eb.bufs.visit
        ea.bufs.visit, eb.bufs.visit = vb, va
        cross["fastest env with the VISIT maps of the slowest"] = timed(ea, 0x30)
        cross["slowest with the visit maps of the fastest"] = timed(eb, 0x30)
        ea.bufs.visit, eb.bufs.visit = va, vb
    print(json.dumps({"variant": variant, "rows": rows, "cross": cross}), flush=True)
    del envs, pads, ring
    torch.cuda.empty_cache()
