#!/usr/bin/env python3
"""What distinguishes a "fast" from a "slow" allocation of the observation buffer?  ONE process, one
1M x 11x11 batch, K separately allocated observation buffers; the step kernel runs RUNS launches on each in turn with the
library's default policy.  Run it directly under rocprofv3 with a --pmc group and --kernel-trace: every dispatch then has
its duration and its counters, and tools/placement_pmc_summary.py groups them by buffer (the dispatch order is fixed:
WARM warm-up launches, then K x RUNS).
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d OUT -- python3 tools/placement_pmc.py [K] [hint]
Without a profiler it prints the per-buffer event times itself."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

WARM, RUNS = 150, 14
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
HINT = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
MODE = sys.argv[3] if len(sys.argv) > 3 else "torch"      # torch: K torch.empty buffers; arena: K slices of ONE allocation, 2 MiB aligned

lm = importlib.import_module("gym-lmaze_amd")
N, G = 1 << 20, 11
env = lm.LmazeVecEnv(N, variant="v0", layout=lm.layouts.open_room(G, (G // 2, G // 2)), seed=1)
env.params.launch_hint = HINT
rows = 80
acts = torch.randint(0, 4, (rows, N), dtype=torch.int32, device="cuda")
nbytes = N * G * G * 4
if MODE == "arena":
    step = (nbytes + (2 << 20) - 1) & ~((2 << 20) - 1)
    arena = torch.empty(K * step + (2 << 20), dtype=torch.uint8, device="cuda")
    off0 = (-arena.data_ptr()) % (2 << 20)
    bufs = [arena[off0 + i * step: off0 + i * step + nbytes].view(torch.int32).view(N, G, G) for i in range(K)]
else:
    bufs = [env.obs] + [torch.empty_like(env.obs) for _ in range(K - 1)]
t = 0
for _ in range(WARM):
    env._launch_step(acts[t % rows].data_ptr(), bufs[0].data_ptr(), False)
    t += 1
torch.cuda.synchronize()
out = []
for i, b in enumerate(bufs):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(RUNS):
        if k == 2:
            e0.record()
        env._launch_step(acts[t % rows].data_ptr(), b.data_ptr(), False)
        t += 1
    e1.record()
    e1.synchronize()
    out.append({"buffer": i, "ptr": hex(b.data_ptr()), "ptr_mod_2MiB": b.data_ptr() % (2 << 20), "us": round(e0.elapsed_time(e1) / (RUNS - 2) * 1e3, 2)})
print(json.dumps({"mode": MODE, "hint": HINT, "warm": WARM, "runs": RUNS, "kernel": lm._abi.describe_step(env.params, N), "buffers": out}), flush=True)
