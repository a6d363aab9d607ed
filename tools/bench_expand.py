#!/usr/bin/env python3
"""Developer bench: the reference-layout render kernels (lmaze_render_expanded, lmaze_expand_planes)."""
import ctypes as C
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    pkg = importlib.import_module("gym-lmaze_amd")
    abi = pkg._abi
    st = torch.cuda.current_stream().cuda_stream
    for (N, G, E, masks) in ((65536, 12, 7, (1, 2, 4, 8)), (65536, 11, 7, (1, 2, 4, 8)), (65536, 18, 4, (8, 1, 4)),
                             (16384, 32, 7, (1, 2, 4, 8)),
                             # x1: the unexpanded float planes a policy network takes (LmazeVecEnv.planes())
                             (1 << 20, 11, 1, (1, 2, 4, 8)), (1 << 20, 8, 1, (1, 2, 4, 8)), (1 << 18, 32, 1, (1, 2, 4, 8))):
        obs = torch.randint(0, 16, (N, G, G), dtype=torch.int32, device="cuda")
        out = torch.empty((N, len(masks), G * E, G * E), dtype=torch.float32, device="cuda")
        m = (C.c_int32 * len(masks))(*masks)

        def run(k):
            for _ in range(k):
                rc = abi.lib.lmaze_render_expanded(obs.data_ptr(), G, E, m, len(masks), out.data_ptr(), N, st)
                assert rc == 0
        run(40)          # the first launches of a process run slow (clocks still ramping)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(20); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        nbytes = out.numel() * 4 + obs.numel() * 4
        print(json.dumps({"kernel": "render_expanded", "N": N, "G": G, "E": E, "C": len(masks), "ms": ms,
                          "GBs": nbytes / ms / 1e6, "frac": nbytes / ms / 1e6 / 8000, "env_per_s": N / ms * 1e3}))
        del obs, out
    for (N, Cn, g, E) in ((1 << 18, 5, 5, 7), (1 << 18, 7, 5, 7), (1 << 18, 4, 5, 7)):
        planes = torch.rand((N, Cn, g, g), dtype=torch.float32, device="cuda")
        out = torch.empty((N, Cn, g * E, g * E), dtype=torch.float32, device="cuda")

        def run(k):
            for _ in range(k):
                rc = abi.lib.lmaze_expand_planes(planes.data_ptr(), Cn, g, E, out.data_ptr(), N, st)
                assert rc == 0
        run(40)          # the first launches of a process run slow (clocks still ramping)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(20); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        nbytes = out.numel() * 4 + planes.numel() * 4
        print(json.dumps({"kernel": "expand_planes", "N": N, "C": Cn, "g": g, "E": E, "ms": ms,
                          "GBs": nbytes / ms / 1e6, "frac": nbytes / ms / 1e6 / 8000, "env_per_s": N / ms * 1e3}))
        del planes, out


if __name__ == "__main__":
    main()
