#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched L-maze step path on N MI355X of one node.

Workload (BASELINE.json configs[2]): 1 048 576 parallel 11x11 mazes per
GPU, v0 transition rules, shared open-room layout, compact int32 observation fully
re-rendered every step, uniform random actions pre-generated on the device.  A "step" is one
lmaze_step_v0 launch over the whole per-GPU batch.  With --gpus N every rank owns its own
1 048 576 envs (weak scaling, configs[3] at N=8); envs are independent, so there is no
collective on the step path -- torch.distributed is used only for the start/stop barrier
and the max-over-ranks time.

One JSON line on rank 0; see README/DESIGN.md for the fields.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def bytes_per_env_step(G, per_env_layout=False):
    """reads action 4 + ball 8 + stepCount 4 + reward 4; writes ball 8 + stepCount 4
    + reward 4 + done 1; obs write 4*G*G; + G*G layout bytes when every env has its own maze."""
    return 37 + 4 * G * G + (G * G if per_env_layout else 0)


def cpu_baseline(G, layout_codes, budget_s=12.0):
    """The C oracle (a port, not the reference interpreter) on this box's host cores, on a
    bounded sample of the same workload."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    N = 1 << 18
    rs = np.random.RandomState(1)
    p = O.params(O.VARIANT_V0, G, O.LAYOUT_SHARED)
    ok = np.argwhere((layout_codes != ord("W")) & (layout_codes != ord("X")))
    ball = np.ascontiguousarray(ok[rs.randint(len(ok), size=N)].astype(np.int32))
    sc = np.zeros(N, np.int32)
    rew = np.zeros(N, np.float32)
    done = np.zeros(N, np.uint8)
    gc = np.zeros(N, np.int32)
    obs = np.zeros((N, G, G), np.int32)
    acts = [rs.randint(0, 4, N).astype(np.int32) for _ in range(8)]
    lay = np.ascontiguousarray(layout_codes)
    # the GPU box gives one GPU's share of the host: 16 cores (gpurun notes); do not grab all 256
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    O.set_threads(threads)
    O.step_v0(p, lay, acts[0], ball, sc, rew, done, gc, obs)   # warm-up / page-in
    t0 = time.perf_counter()
    O.step_v0(p, lay, acts[1], ball, sc, rew, done, gc, obs)
    one = max(time.perf_counter() - t0, 1e-6)
    steps = int(max(4, min(20000, budget_s / one)))
    t0 = time.perf_counter()
    for t in range(steps):
        O.step_v0(p, lay, acts[t & 7], ball, sc, rew, done, gc, obs)
    dt = time.perf_counter() - t0
    return {"value": N * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": "%d envs x %d steps of the same %dx%d v0 workload, C oracle (OpenMP, %d threads), %.1f s"
                      % (N, steps, G, G, threads, dt)}


# algorithmic HBM bytes per env-step of the foveal variants: per-env scalars read + written,
# the float32 [C,5,5] observation written, and for v4 the 18x18 float32 visit map read + written
FOVEAL_BYTES = {"v1": 28 + 26 + 400, "v2": 28 + 17 + 500, "v4": 28 + 17 + 2 * 1296 + 700}
FOVEAL_ACTIONS = {"v1": 4, "v2": 25, "v4": 25}


def cpu_baseline_foveal(variant, budget_s=12.0):
    """The C oracle's foveal step on this box's host cores, bounded sample of the same workload."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    pkg = importlib.import_module("gym-lmaze_amd")
    vid = {"v1": O.VARIANT_V1, "v2": O.VARIANT_V2, "v4": O.VARIANT_V4}[variant]
    tabs = [pkg.layouts.to_codes(t) for t in ((pkg.layouts.V1_GRID_14,) if variant == "v1" else pkg.layouts.FOVEAL_GRIDS_18)]
    layouts = np.ascontiguousarray(np.stack(tabs))
    G, N = layouts.shape[-1], 1 << 17
    p = O.foveal_params(vid, G, len(tabs))
    st = O.FovealState(vid, N, G)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    O.set_threads(threads)
    O.foveal_reset(p, layouts, None, 1, 1, 0, st)
    rs = np.random.RandomState(1)
    acts = [rs.randint(0, FOVEAL_ACTIONS[variant], N).astype(np.int32) for _ in range(8)]
    O.foveal_step(p, layouts, acts[0], st)
    t0 = time.perf_counter()
    O.foveal_step(p, layouts, acts[1], st)
    one = max(time.perf_counter() - t0, 1e-6)
    steps = int(max(4, min(20000, budget_s / one)))
    t0 = time.perf_counter()
    for t in range(steps):
        O.foveal_step(p, layouts, acts[t & 7], st)
    dt = time.perf_counter() - t0
    return {"value": N * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": "%d envs x %d steps of the same lmaze-%s workload, C oracle (OpenMP, %d threads), %.1f s"
                      % (N, steps, variant, threads, dt)}


def measured_ceiling(pkg, nbytes, dev, reps=20):
    """The box's own write / copy ceilings: lmaze_bandwidth_probe over a scratch buffer the size
    of the obs buffer, events on the launch stream.  Reported beside the 8 TB/s peak, never instead of it."""
    import torch
    abi = importlib.import_module(pkg.__name__ + "._abi")
    nbytes = min(int(nbytes), 4 << 30) & ~15
    src = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream(dev).cuda_stream
    out = {"unit": "GB/s", "bytes": nbytes}
    with torch.cuda.device(dev):
        for name, s_ptr, moved in (("fill", None, nbytes), ("copy", src.data_ptr(), 2 * nbytes)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(reps + 5):
                if i == 5:
                    e0.record()
                abi.check("lmaze_bandwidth_probe", abi.lib.lmaze_bandwidth_probe(s_ptr, dst.data_ptr(), nbytes, st))
            e1.record()
            torch.cuda.synchronize()
            out[name] = moved / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", choices=["c3", "c2", "c5", "v1", "v2", "v4"], default="c3",
                    help="c3 (default, the metric's config): 1 048 576 x 11x11 shared layout; "
                         "c2: 65 536 x 8x8; c5: 1 048 576 x 32x32 with per-env random layouts; "
                         "v1 / v2 / v4: 1 048 576 envs of the foveal variants (5x5 window observations; SURVEY 8(f)3)")
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (overrides the workload's)")
    ap.add_argument("--grid", type=int, default=None)
    ap.add_argument("--per-env-layouts", action="store_true", help="own random maze per env")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0,
                    help="CPU work spent on the cpu_baseline sample (the C oracle on the host cores)")
    ap.add_argument("--auto-reset", action="store_true",
                    help="lmaze_step_v0_autoreset: done envs are re-placed inside the step kernel (rollouts > 1 episode)")
    ap.add_argument("--no-autotune", action="store_true",
                    help="keep the library's default launch policy instead of LmazeVecEnv.autotune()")
    ap.add_argument("--launch-hint", type=int, default=None,
                    help="fixed LmazeParams.launch_hint (workgroups per CU), skipping the autotune; used for the "
                         "rocprofv3 passes so that every profiled launch runs the policy the bench line was measured with")
    ap.add_argument("--graph", action="store_true",
                    help="capture the K timed launches into one hipGraph and time its replay (launch-bound sizes)")
    ap.add_argument("--action-rows", type=int, default=None,
                    help="rows of the pre-generated action tensor int32[rows, N] (default: one per timed step, the "
                         "[T,N] tensor of SURVEY 8(d) C3, capped at 1024 rows; fewer rows are cycled -- 32 rows of "
                         "1M envs are 134 MB and stay in the Infinity Cache, which flatters the step kernel)")
    args = ap.parse_args()

    import numpy as np
    import torch

    foveal = args.workload in FOVEAL_BYTES
    preset = {"c3": (1 << 20, 11, False), "c2": (65536, 8, False), "c5": (1 << 20, 32, True),
              "v1": (1 << 20, 14, False), "v2": (1 << 20, 18, False), "v4": (1 << 20, 18, False)}[args.workload]
    if foveal and (args.graph or args.per_env_layouts or args.grid is not None):
        raise SystemExit("--graph / --per-env-layouts / --grid do not apply to the foveal workloads")
    args.envs = args.envs if args.envs is not None else preset[0]
    args.grid = args.grid if args.grid is not None else preset[1]
    args.per_env_layouts = args.per_env_layouts or preset[2]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d"
                             % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernels are the only step path")
    # LMAZE_BENCH_BACKEND=gloo is a rehearsal switch: several ranks share the visible GPU(s) and the barrier /
    # MAX go over gloo, to exercise the multi-rank code path on a one-GPU box.  Never used for reported numbers.
    backend = os.environ.get("LMAZE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a banner on fd 1 when it loads; keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm; barrier + one MAX only
            else:
                dist.init_process_group(backend)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    pkg = importlib.import_module("gym-lmaze_amd")
    G, N = args.grid, args.envs
    env_base = rank * N
    tuned, layout = None, None
    R = args.action_rows if args.action_rows else max(1, min(args.steps, 1024))
    gen = torch.Generator(device=dev).manual_seed(1 + rank)      # torch's device generator is Philox
    if foveal:
        variant = args.workload
        env = pkg.LmazeFovealVecEnv(N, variant=variant, device=dev, seed=1, env_base=env_base)
        G = env.grid
        if variant == "v1":        # the two-level loop's upper half: a foveal goal per env (lmaze_env_v1.py:104-110)
            env.set_foveal_goal(torch.randint(0, 5, (N, 2), dtype=torch.int32, device=dev, generator=gen))
        workload = ("%d x lmaze-%s per GPU (%dx%d layouts, 5x5 window, float32 [%d,5,5] observation%s)"
                    % (N, variant, G, G, env.channels, ", float32 visit map" if variant == "v4" else ""))
        actions = torch.randint(0, FOVEAL_ACTIONS[variant], (R, N), dtype=torch.int32, device=dev, generator=gen)
        # warm the device as LmazeVecEnv.autotune() does for the grid workloads (cold clocks)
        for t in range(150):
            env.step(actions[t % R], auto_reset=args.auto_reset)

        def run(k0, k, captured=False):
            for t in range(k0, k0 + k):
                env.step(actions[t % R], auto_reset=args.auto_reset)
    else:
        if args.workload == "c2":
            layout = pkg.layouts.to_codes(pkg.layouts.GRID_8_BORDERED)   # lmaze_env.py:28-35 literal, bordered
        else:
            layout = pkg.layouts.to_codes(pkg.layouts.open_room(G, (G // 2, G // 2)))
        if args.per_env_layouts:
            lgen = torch.Generator(device=dev).manual_seed(7 + rank)
            lay = torch.where(torch.rand((N, G, G), device=dev, generator=lgen) < 0.25, ord("W"), ord("B")).to(torch.uint8)
            lay[:, 0, :] = ord("W"); lay[:, -1, :] = ord("W"); lay[:, :, 0] = ord("W"); lay[:, :, -1] = ord("W")
            lay[:, 1, 1] = ord("S")
            lay[:, G - 2, G - 2] = ord("X")
            env = pkg.LmazeVecEnv(N, variant="v0", per_env_layouts=lay, device=dev, seed=1, env_base=env_base)
            workload = "%d x %dx%d mazes per GPU, v0 rules, per-env random layouts (p_wall 0.25), compact int32 obs" % (N, G, G)
        else:
            env = pkg.LmazeVecEnv(N, variant="v0", layout=layout, device=dev, seed=1, env_base=env_base)
            workload = "%d x %dx%d mazes per GPU, v0 rules, shared open-room layout, compact int32 obs" % (N, G, G)

        actions = torch.randint(0, 4, (R, N), dtype=torch.int32, device=dev, generator=gen)
        row_ptr = [actions[r].data_ptr() for r in range(R)]
        env._tuner = None           # the bench picks the policy before the timed region, never during it
        if args.launch_hint is not None:
            env.params.launch_hint = args.launch_hint
        elif not args.no_autotune:
            # untimed: picks (workgroups per CU, chunks per workgroup) for this shape and device, on the very
            # action tensor the timed steps read (cache-resident or not decides the ranking)
            tuned = env.autotune(auto_reset=args.auto_reset, actions=actions)

        def run(k0, k, captured=False):
            for t in range(k0, k0 + k):
                # under capture the reset epoch is a device word handed from launch to launch (slot = launch index)
                env.step_raw(row_ptr[t % R], auto_reset=args.auto_reset,
                             epoch_slot=(t - k0) if (captured and args.auto_reset) else None)

    with torch.cuda.device(dev):
        run(0, args.warmup)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        # HIP events on the stream the kernels are launched on (torch's current stream)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        graph = None
        if args.graph:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    run(args.warmup, args.steps, captured=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        ev0.record()
        if graph is not None:
            if args.auto_reset:
                env.begin_replay(args.steps)
            graph.replay()
        else:
            run(args.warmup, args.steps)
        ev1.record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
            elapsed = pkg.max_over_ranks(elapsed, device=dev)
    # per-launch GPU time from HIP events recorded on the launch stream
    kern_ms = float(ev0.elapsed_time(ev1) / args.steps)   # ms per launch, launch gaps included

    # sanity: the run really stepped (every env advanced warmup+steps times)
    if foveal:      # like the reference, stepping goes on past `done` unless the reset is fused in
        top = int(env.step_count.max().item())
        assert top >= 1 and (not args.auto_reset or top <= env.params.step_limit + 1)
    elif not args.auto_reset:
        assert int(env.step_count.min().item()) == args.warmup + args.steps
    else:  # episodes restart: nobody is past the step limit, and everybody moved
        assert 1 <= int(env.step_count.min().item()) and int(env.step_count.max().item()) <= env.step_limit

    ceiling = measured_ceiling(pkg, env.obs.numel() * 4, dev) if rank == 0 else None

    if rank == 0:
        B = FOVEAL_BYTES[args.workload] if foveal else bytes_per_env_step(G, args.per_env_layouts)
        total_steps = world * N * args.steps
        value = total_steps / elapsed
        achieved = N * B / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = args.workload if foveal else "g%d_%s" % (G, "perenv" if args.per_env_layouts else "shared")
                rec = tj.get(key, {})
                # the PMC passes profiled one launch shape: only quote them for that shape
                if rec.get("algorithmic_bytes_per_launch") == N * B:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env steps/sec (whole node), 1M parallel 11x11 mazes at 1/2/4/8 MI355X" if args.workload == "c3" and N == (1 << 20)
                      else "env steps/sec (whole node); workload '%s', NOT the configuration BASELINE.json's metric is quoted on" % args.workload,
            "workload_id": args.workload,
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if foveal else "int32", "data": "synthetic",
            "config": {"workload": workload, "envs_per_gpu": N, "grid": G, "global_envs": world * N,
                       "parallelism": "independent env shards, no collective on the step path",
                       "actions": "uniform{0..%d} int32[%d,N] on the device, row t %% rows at step t, torch Philox seed 1+rank"
                                  % ((FOVEAL_ACTIONS[args.workload] if foveal else 4) - 1, R),
                       "auto_reset": bool(args.auto_reset), "hip_graph": bool(args.graph),
                       "collective_backend": ("rccl" if backend == "nccl" else backend + " (REHEARSAL, ranks share a GPU)")
                       if dist is not None else None,
                       "launch_hint": None if foveal else int(env.params.launch_hint),
                       "autotune_ms": {("%dx%d" % k if isinstance(k, tuple) else str(k)): round(v, 5)
                                       for k, v in (tuned or {}).items()}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("lmaze::foveal_kernel<%s, FM_STEP>" % args.workload) if foveal else
                                   "lmaze::step_%s_kernel<%d, v0>" % ("perenv" if args.per_env_layouts else "shared", G),
                         "bytes_per_env_step": B, "kernel_ms_avg": kern_ms, "measured_ceiling": ceiling},
        }
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N=1 only
            out["cpu_baseline"] = (cpu_baseline_foveal(args.workload, args.cpu_baseline_seconds) if foveal
                                   else cpu_baseline(G, layout, args.cpu_baseline_seconds))
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
