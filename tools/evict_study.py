#!/usr/bin/env python3
"""What a step costs when something else runs between two steps (a policy network, say) and evicts the per-env
state and the actions from the caches: a 512-MB copy between the steps stands in for it.  Prints the step time
with and without the interloper (its own time measured alone and subtracted)."""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("gym-lmaze_amd")
abi = pkg._abi

N, G, T = 1 << 20, 11, 300
env = pkg.LmazeVecEnv(N, variant="v0", layout=pkg.layouts.to_codes(pkg.layouts.open_room(G, (5, 5))), online_autotune=False)
acts = torch.randint(0, 4, (T, N), dtype=torch.int32, device="cuda")
src = torch.empty(64 << 20, dtype=torch.int32, device="cuda")     # 256 MB
dst = torch.empty_like(src)
st = torch.cuda.current_stream().cuda_stream
nbytes = src.numel() * 4


WRITES = os.environ.get("INTERLOPER", "read") == "copy"
SPIN = os.environ.get("INTERLOPER", "read") == "spin"      # no memory traffic at all: only separates the launches


def interloper():
    """read: 512 MB of clean lines pass through the caches.  copy: 256 MB read + 256 MB written -- the dirty half
    is written back to HBM while the next step runs, traffic this script then charges to the step."""
    if SPIN:
        torch.cuda._sleep(200000)        # ~100 us of spinning on one wave
    elif WRITES:
        abi.lib.lmaze_bandwidth_probe(src.data_ptr(), dst.data_ptr(), nbytes, st)
    else:
        src.sum(); dst.sum()


def timed(fn, reps):
    fn(20)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(reps); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for _ in range(200):
    env.step_raw(acts[0].data_ptr())
TOUCH = os.environ.get("TOUCH_INPUTS", "0") == "1"   # after the interloper, read state + action row back into the caches


def step_events(k, with_interloper):
    """Mean duration of the step launches alone (an event pair around every one of them)."""
    pairs = []
    for t in range(k):
        if with_interloper:
            interloper()
            if TOUCH:
                env._state.sum(); acts[t % T].sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.step_raw(acts[t % T].data_ptr())
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs[20:])
    return d[len(d) // 2]


for hint in [int(h, 0) for h in (sys.argv[1:] or ["0x23", "0x13", "0x25", "0x18", "0x24"])]:
    env.params.launch_hint = hint
    alone = timed(lambda k: [env.step_raw(acts[t % T].data_ptr()) for t in range(k)], 300)
    print(json.dumps({"hint": hex(hint), "step_us_back_to_back": round(alone, 2),
                      "step_us_event_pairs_back_to_back": round(step_events(200, False), 2),
                      "step_us_event_pairs_after_interloper": round(step_events(200, True), 2)}), flush=True)
