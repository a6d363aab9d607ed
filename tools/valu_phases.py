#!/usr/bin/env python3
"""Instruction counts of a foveal step per phase, for `rocprofv3 --pmc SQ_INSTS_VALU ...`: the experiment build
(LMAZE_HIP_LIB=tools/_exp/liblmaze_hip_exp.so) with one phase switched off through launch_hint bits 16+.
200 warm steps run at another envs-per-workgroup (a different template instantiation, so the counters of the
measured launches group under their own kernel name), then 8 launches with the switch.
    python tools/valu_phases.py v5 <base hint> <switch bits>"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("gym-lmaze_amd")
variant, base, xp = sys.argv[1], int(sys.argv[2], 0), int(sys.argv[3], 0)
N, R = 1 << 20, 16
env = pkg.LmazeFovealVecEnv(N, variant=variant, seed=1)
hi = 4 if variant in ("v1", "v5") else 25
acts = torch.randint(0, hi, (R, N), dtype=torch.int32, device="cuda")
goals = torch.randint(0, 25, (R, N), dtype=torch.int32, device="cuda")
if variant == "v1":
    env.set_foveal_goal(torch.randint(0, 5, (N, 2), dtype=torch.int32, device="cuda"))
if variant == "v5":
    env.foveal_done.fill_(True)


def run(k, t0=0):
    for t in range(t0, t0 + k):
        if variant == "v5":
            env.hier_step_raw(acts[t % R].data_ptr(), goals[t % R].data_ptr())
        else:
            env.step_raw(acts[t % R].data_ptr())


warm = (base & ~0xF0) | (0x30 if (base & 0xF0) != 0x30 else 0x40)
env.params.launch_hint = warm
run(200)
env.params.launch_hint = (xp << 16) | base
run(8, 200)
torch.cuda.synchronize()
print("done", variant, hex(base), xp)
