#!/usr/bin/env python3
"""Developer bench for the foveal variants (not the driver's metric; bench.py is).
python tools/bench_foveal.py [--envs N] [--steps K]  -> one JSON line per variant."""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per env-step: per-env scalars read + written, visit map
# read + written (v4: every step; v5: whole plane only on localDone, else the two sampled windows), obs
BYTES = {"v1": 28 + 26 + 400, "v2": 28 + 17 + 500, "v4": 28 + 17 + 2 * 1296 + 700}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("gym-lmaze_amd")
    N = args.envs
    # warm the device first: the first ~100 launches of a process run 10-40 % slow (clocks still ramping) and
    # would be charged to whichever variant is measured first
    w = pkg.LmazeVecEnv(N, variant="v0", layout=pkg.layouts.to_codes(pkg.layouts.open_room(11, (5, 5))), online_autotune=False)
    a0 = torch.randint(0, 4, (N,), dtype=torch.int32, device="cuda")
    for _ in range(400):
        w.step_raw(a0.data_ptr())
    torch.cuda.synchronize()
    del w, a0
    for variant in ("v1", "v2", "v4", "v5"):
        env = pkg.LmazeFovealVecEnv(N, variant=variant, seed=1)
        hi = 4 if variant in ("v1", "v5") else 25
        acts = torch.randint(0, hi, (16, N), dtype=torch.int32, device="cuda")
        goals = torch.randint(0, 25, (N,), dtype=torch.int32, device="cuda")
        if variant == "v1":
            env.set_foveal_goal(torch.randint(0, 5, (N, 2), dtype=torch.int32, device="cuda"))

        def run(k):
            for t in range(k):
                if variant == "v5" and t % 10 == 0:
                    env.planner_step(goals, mask=env.foveal_done if t else None)
                env.step(acts[t % 16])

        run(args.warmup)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(args.steps)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        out = {"variant": variant, "envs": N, "ms_per_step": ms, "env_steps_per_s": N / (ms * 1e-3)}
        if variant in BYTES:
            out["bytes_per_env_step"] = BYTES[variant]
            out["achieved_GBs"] = N * BYTES[variant] / (ms * 1e-3) / 1e9
            out["frac_of_8TBs"] = out["achieved_GBs"] / 8000.0
        else:
            ld = float(env.foveal_done.float().mean().item())
            out["local_done_fraction"] = ld
        print(json.dumps(out))
        del env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
