#!/usr/bin/env python3
"""Where a foveal step launch spends its time: the experimental build (`make -C gym-lmaze_amd/csrc experiment`: csrc
compiled with -DLMAZE_EXPERIMENT into tools/_exp/liblmaze_hip_exp.so, selected through LMAZE_HIP_LIB) can switch off the per-workgroup set-up (bit 16 of
launch_hint), the observation stores (bit 18) and phase 1 (bit 19).  Results are garbage in those modes; only the
time counts.   LMAZE_HIP_LIB=tools/_exp/liblmaze_hip_exp.so python tools/foveal_decompose.py v2 v1"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("gym-lmaze_amd")
N, K = 1 << 20, 60
modes = {"full": 0, "no_setup": 1, "no_stores": 4, "no_phase1": 8, "stores_only": 9, "phase1 only": 5, "setup only": 12}
if os.environ.get("DECOMPOSE_VISIT"):     # v4 / v5: the visit-map stream (bit 13) as well
    modes = {"full": 0, "no_setup": 1, "no_stores": 4, "no_phase1": 8, "no_visit_stream": 32, "stores_only": 41,
             "visit_stream only": 13, "phase1 only": 37, "phase1 + visit": 5}
if os.environ.get("DECOMPOSE_R3"):        # round 3: the gathered-tile visit phase: per-cell work off (256), "previous" window tiles off (512)
    modes = {"full": 0, "no cell work": 256, "no prev tiles": 512, "neither": 768, "no_visit": 32, "no_stores": 4,
             "no_stores no cell work": 260, "no_stores neither": 772}
if os.environ.get("DECOMPOSE_NTMAP"):     # v4: non-temporal accesses for the visit-map stream
    modes = {"full": 0, "nt map stores": 64, "nt map loads": 128, "nt both": 192, "visit only": 13, "visit only nt both": 13 + 192,
             "visit only nt stores": 13 + 64}
if os.environ.get("DECOMPOSE_PLAIN"):     # plain against non-temporal observation stores, whole kernel, caps x sizes
    modes = {"full": 0, "full plain-stores": 2}
if os.environ.get("DECOMPOSE_CAPS"):      # second study: does capping the resident workgroups help the bare store stream?
    modes = {"full": 0, "stores_only": 9, "full interleaved-pieces": 16, "stores_only interleaved-pieces": 25,
             "stores_only interleaved-pieces plain": 27}
CAPS = (0, 3, 4, 5, 6) if (os.environ.get("DECOMPOSE_CAPS") or os.environ.get("DECOMPOSE_PLAIN")) else (0,)
for variant in (sys.argv[1:] or ["v2", "v1"]):
    env = pkg.LmazeFovealVecEnv(N, variant=variant, seed=1)
    hi = 4 if variant in ("v1", "v5") else 25
    R = 80
    acts = torch.randint(0, hi, (R, N), dtype=torch.int32, device="cuda")
    if variant == "v1":
        env.set_foveal_goal(torch.randint(0, 5, (N, 2), dtype=torch.int32, device="cuda"))
    ap = [acts[r].data_ptr() for r in range(R)]
    t = 0

    goals = torch.randint(0, 25, (R, N), dtype=torch.int32, device="cuda")
    gp = [goals[r].data_ptr() for r in range(R)]
    if variant == "v5":
        env.foveal_done.fill_(True)

    def run(k):
        global t
        for _ in range(k):
            if variant == "v5":
                env.hier_step_raw(ap[t % R], gp[t % R])
            else:
                env.step_raw(ap[t % R])
            t += 1

    run(200)
    snap = env._state.clone()
    out = {}
    for rnd in range(2):
        for epb, name, cap in [(e, m, c) for e in ((4,) if os.environ.get('DECOMPOSE_CAPS') else ((3,) if os.environ.get('DECOMPOSE_NTMAP') else ((3, 4, 5) if os.environ.get('DECOMPOSE_PLAIN') else (2, 3, 4)))) for m in modes for c in CAPS]:
            if True:
                xp = modes[name]
                env._state.copy_(snap)
                env.params.launch_hint = (xp << 16) | (epb << 4) | cap
                run(3)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(K)
                e1.record()
                e1.synchronize()
                key = "epb%d %s%s" % (16 << (epb - 1), name, " cap%d" % cap if cap else "")
                ms = e0.elapsed_time(e1) / K
                out[key] = min(ms, out.get(key, ms))
    print(json.dumps({"variant": variant, "us": {k: round(v * 1e3, 1) for k, v in out.items()}}), flush=True)
