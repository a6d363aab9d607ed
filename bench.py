#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched L-maze step path on N MI355X of one node.

Workload (BASELINE.json configs[2]): 1 048 576 parallel 11x11 mazes per
GPU, v0 transition rules, shared open-room layout, compact int32 observation fully
re-rendered every step, uniform random actions pre-generated on the device.  A "step" is one
lmaze_step_v0 launch over the whole per-GPU batch.  With --gpus N every rank owns its own
1 048 576 envs (weak scaling, configs[3] at N=8); envs are independent, so there is no
collective on the step path -- torch.distributed is used only for the start/stop barrier,
the max-over-ranks time and a gather of the per-rank times.

Launching: `python bench.py --gpus N` from a plain shell starts its own N ranks (one child process per GPU,
spawned before anything in the parent touches torch or the GPU); under `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N` the ranks are already there and each process is one of them.

One JSON line on rank 0; see README/DESIGN.md for the fields.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)

# The reference's own interpreter, 1 core of the build container's Xeon @ 2.1 GHz: it
# cannot travel to the GPU box, so these are quoted, not measured by this run.
REFERENCE_INTERPRETER_STEPS_PER_S = {"c3": 85.0, "c2": 153.0, "c5": 10.0, "v1": 370.0, "v2": 345.0, "v4": 263.0,
                                     "v5": 160.0}


def bytes_per_env_step(G, per_env_layout=False):
    """reads action 4 + ball 8 + stepCount 4 + reward 4; writes ball 8 + stepCount 4
    + reward 4 + done 1; obs write 4*G*G; + G*G layout bytes when every env has its own maze."""
    return 37 + 4 * G * G + (G * G if per_env_layout else 0)


def _timed_loop(fn, budget_s, cap=20000):
    """Call fn(t) until about budget_s seconds have gone by (after one untimed warm-up call); returns (calls, seconds)."""
    fn(0)                                              # warm-up / page-in
    t0 = time.perf_counter()
    steps = 0
    while True:
        fn(steps)
        steps += 1
        dt = time.perf_counter() - t0
        if (dt >= budget_s and steps >= 3) or steps >= cap:
            return steps, dt


def host_threads():
    # the GPU box gives one GPU's share of the host: 16 cores (gpurun notes); do not grab all 256
    return max(1, min(16, len(os.sched_getaffinity(0))))


def cpu_baseline(workload, G, layout_codes, budget_s=12.0):
    """CPU legs beside the GPU number, all on a bounded sample of the same 11x11 / 8x8 / 32x32 v0
    workload: the C oracle (a port, not the reference interpreter) on this box's host cores (`value`), the same
    on one thread, and the NumPy-vectorised restatement; the reference interpreter's own figure is quoted from
    BASELINE.md (it cannot travel here)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    import oracle_numpy as ON
    N = 1 << 18
    rs = np.random.RandomState(1)
    p = O.params(O.VARIANT_V0, G, O.LAYOUT_SHARED)
    ok = np.argwhere((layout_codes != ord("W")) & (layout_codes != ord("X")))
    lay = np.ascontiguousarray(layout_codes)
    threads = host_threads()

    def state(n):
        return dict(ball=np.ascontiguousarray(ok[rs.randint(len(ok), size=n)].astype(np.int32)), sc=np.zeros(n, np.int32),
                    rew=np.zeros(n, np.float32), done=np.zeros(n, np.uint8), gc=np.zeros(n, np.int32),
                    obs=np.zeros((n, G, G), np.int32), acts=[rs.randint(0, 4, n).astype(np.int32) for _ in range(8)])

    def c_leg(n, nthreads, seconds):
        s = state(n)
        O.set_threads(nthreads)
        pp, ll, what = p, lay, "shared layout"
        if workload == "c5":
            # per-env random mazes as the GPU workload has them, the ball placed by the oracle's reset
            n = min(n, 1 << 15)
            s = state(n)
            ll = np.where(rs.rand(n, G, G) < 0.25, ord("W"), ord("B")).astype(np.uint8)
            ll[:, 0, :] = ll[:, -1, :] = ll[:, :, 0] = ll[:, :, -1] = ord("W")
            free = np.where(ll.reshape(n, -1) == ord("B"), rs.rand(n, G * G), -1.0)
            ll.reshape(n, -1)[np.arange(n), free.argmax(1)] = ord("X")
            pp, what = O.params(O.VARIANT_V0, G, O.LAYOUT_PER_ENV), "per-env random layouts"
            O.reset(pp, ll, None, 7, 0, s["ball"], None, s["sc"], s["rew"], s["done"], None)
        steps, dt = _timed_loop(lambda t: O.step_v0(pp, ll, s["acts"][t & 7], s["ball"], s["sc"], s["rew"], s["done"],
                                                    s["gc"], s["obs"]), seconds)
        return n * steps / dt, "%d envs x %d steps of the same %dx%d v0 workload (%s), C oracle (OpenMP, %d thread%s), %.1f s" % (
            n, steps, G, G, what, nthreads, "" if nthreads == 1 else "s", dt)

    v, sample = c_leg(N, threads, 0.6 * budget_s)
    out = {"value": v, "unit": "env-steps/s", "cores": threads, "kind": "port", "sample": sample}
    v1, s1 = c_leg(N >> 2, 1, 0.2 * budget_s)
    out["single_thread"] = {"value": v1, "unit": "env-steps/s", "cores": 1, "kind": "port", "sample": s1}
    s = state(N >> 2)
    static = ON.static_bits(lay)
    steps, dt = _timed_loop(lambda t: ON.step_v0(lay, static, s["acts"][t & 7], s["ball"], s["sc"], s["rew"], s["done"],
                                                 s["gc"], s["obs"]), 0.2 * budget_s)
    out["numpy_vectorised"] = {"value": (N >> 2) * steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
                               "sample": "%d envs x %d steps, NumPy restatement (oracle/oracle_numpy.py), %.1f s" % (N >> 2, steps, dt)}
    out["reference_interpreter"] = reference_interpreter(workload)
    O.set_threads(threads)
    return out


def reference_interpreter(workload):
    return {"value": REFERENCE_INTERPRETER_STEPS_PER_S.get(workload), "unit": "env-steps/s", "cores": 1,
            "kind": "reference",
            "sample": "the reference's own step() incl. its x7 render loop, BASELINE.md section 2: build container, "
                      "1 core of a Xeon @ 2.1 GHz, NOT this box (the reference cannot travel to the GPU box)"}


# algorithmic HBM bytes per env-step of the foveal variants: per-env scalars read + written and the
# float32 [C,5,5] observation written.  v4 (round 3: clock-relative visit map, a step touches only the window): scalars
# 32 read (action 4, ball 8, step count 4, layout id 4, goal 8, visit clock 4) + 21 written (ball 8, step count 4, reward
# 4, done 1, visit clock 4), observation 700, the 25 cells of the current window read + written (200) and the cells the
# "previous" window shows beside them, read once each -- COUNTED over the timed steps (V4_PREV_ONLY_CELL per cell), as the
# v5 events are; with --auto-reset a reset writes the zeroed map (4 G^2) and the placement (V4_RESET).  Round 2 streamed
# the whole map twice: 28 + 17 + 700 + 2 x 1296 = 3 337.
FOVEAL_BYTES = {"v1": 28 + 26 + 400, "v2": 28 + 17 + 500, "v4": 32 + 21 + 700 + 200}
V4_PREV_ONLY_CELL, V4_RESET = 4, 12 + 1296
FOVEAL_ACTIONS = {"v1": 4, "v2": 25, "v4": 25, "v5": 4}
# v5 two-level step (lmaze_v5_hier_step).  Every env-step: reads action 4 + planner goal 4 + done 1 +
# localDone 1 + ball 8 + layout id 4 + goal 8 + foveal goal 4 + fovea 16 + previous ball 8 + last window 8 + foveal
# goal cell 8 + two step counts 8 + visit clock 4 = 86; writes ball 8 + previous ball 8 + fovea_0 8 + last window 8 + step
# count 4 + two rewards 8 + two done flags 2 = 46; both observations written, 700 + 400; the 25 cells of the current
# window and the 25 values of the "previous" one read, 200.  Per event: plannerStep (the env entered with localDone or
# done) writes 24; reset (entered with done) writes 12 + the zeroed visit map 1296 + the previous-window record 100; an
# env that ends the step with localDone writes the 25 updated cells, its previous-window record and its clock, 204.
# (Round 2: 2 x 1296 per localDone and a 200-byte gather otherwise, 1 705 B per env-step on the same events.)
V5_BASE, V5_PLAN, V5_RESET, V5_UPDATE = 86 + 46 + 1100 + 200, 24, 12 + 1296 + 100, 204


def cpu_baseline_foveal(variant, budget_s=12.0):
    """The C oracle's foveal step on this box's host cores, bounded sample of the same workload."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    pkg = importlib.import_module("gym-lmaze_amd")
    vid = {"v1": O.VARIANT_V1, "v2": O.VARIANT_V2, "v4": O.VARIANT_V4, "v5": O.VARIANT_V5}[variant]
    tabs = [pkg.layouts.to_codes(t) for t in ((pkg.layouts.V1_GRID_14,) if variant == "v1" else pkg.layouts.FOVEAL_GRIDS_18)]
    layouts = np.ascontiguousarray(np.stack(tabs))
    G, N = layouts.shape[-1], 1 << 17
    p = O.foveal_params(vid, G, len(tabs))
    rs = np.random.RandomState(1)

    def leg(n, nthreads, seconds):
        st = O.FovealState(vid, n, G)
        O.set_threads(nthreads)
        acts = [rs.randint(0, FOVEAL_ACTIONS[variant], n).astype(np.int32) for _ in range(8)]
        if variant == "v5":
            goals = [rs.randint(0, 25, n).astype(np.int32) for _ in range(8)]
            O.v5_reset(p, layouts, None, 1, 1, 0, st)
            st.foveal_done[:] = 1                # every env starts with a plannerStep, as the two-level loop does
            fn = lambda t: O.v5_hier_step(p, layouts, acts[t & 7], goals[t & 7], 1, 1 + t, st)   # noqa: E731
        else:
            O.foveal_reset(p, layouts, None, 1, 1, 0, st)
            fn = lambda t: O.foveal_step(p, layouts, acts[t & 7], st)                           # noqa: E731
        steps, dt = _timed_loop(fn, seconds)
        return n * steps / dt, "%d envs x %d steps of the same lmaze-%s workload, C oracle (OpenMP, %d thread%s), %.1f s" % (
            n, steps, variant, nthreads, "" if nthreads == 1 else "s", dt)

    threads = host_threads()
    v, sample = leg(N, threads, 0.75 * budget_s)
    out = {"value": v, "unit": "env-steps/s", "cores": threads, "kind": "port", "sample": sample}
    v1, s1 = leg(N >> 3, 1, 0.25 * budget_s)
    out["single_thread"] = {"value": v1, "unit": "env-steps/s", "cores": 1, "kind": "port", "sample": s1}
    out["reference_interpreter"] = reference_interpreter(variant)
    O.set_threads(threads)
    return out


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


def warm_device(pkg, nbytes, dev, launches=400):
    """`launches` fill-probe launches over a scratch buffer of the observation's size (about 30 ms at 1M x 11x11)."""
    import torch
    abi = importlib.import_module(pkg.__name__ + "._abi")
    nbytes = min(int(nbytes), 2 << 30) & ~15
    dst = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        for _ in range(launches):
            abi.check("lmaze_bandwidth_probe", abi.lib.lmaze_bandwidth_probe(None, dst.data_ptr(), nbytes, st))
        torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` from a plain shell
# ------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(cmd, n, env=None, poll_s=0.05):
    """Start `cmd` n times as ranks 0..n-1 of one node (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in
    the environment, rendezvous on 127.0.0.1) and wait for all of them.  The caller has not touched torch or the
    GPU: the children are fresh processes (never an exec of a process that has initialised HIP).  Rank 0 keeps this
    process's stdout -- the ONE JSON line --, the others' stdout goes to stderr.  If a rank fails the others are
    stopped (exactly the PIDs started here) and its exit code is returned."""
    base = dict(os.environ if env is None else env)
    base.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                 "MASTER_PORT": str(free_port()), "LMAZE_BENCH_SELF_LAUNCHED": "1"})
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL needs it)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(poll_s)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:          # a rank died: the others would wait in a barrier for ever
                    q.terminate()
    for p in procs:
        if p.poll() is None:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", choices=["c3", "c2", "c5", "v1", "v2", "v4", "v5"], default="c3",
                    help="c3 (default, the metric's config): 1 048 576 x 11x11 shared layout; "
                         "c2: 65 536 x 8x8; c5: 1 048 576 x 32x32 with per-env random layouts; "
                         "v1 / v2 / v4: 1 048 576 envs of the foveal variants (5x5 window observations; SURVEY 8(f)3); "
                         "v5: the two-level loop of lmaze-v5 (reset on globalDone, plannerStep on localDone, step) as one "
                         "launch per env-step")
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (overrides the workload's)")
    ap.add_argument("--grid", type=int, default=None)
    ap.add_argument("--per-env-layouts", action="store_true", help="own random maze per env")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0,
                    help="CPU work spent on the cpu_baseline legs (the C oracle on the host cores, on one thread, NumPy)")
    ap.add_argument("--auto-reset", action="store_true",
                    help="lmaze_step_v0_autoreset: done envs are re-placed inside the step kernel (rollouts > 1 episode)")
    ap.add_argument("--no-autotune", action="store_true",
                    help="keep the library's default launch policy instead of LmazeVecEnv.autotune()")
    ap.add_argument("--launch-hint", type=int, default=None,
                    help="fixed launch_hint (LmazeParams / LmazeFovealParams), skipping the autotune; used for the "
                         "rocprofv3 passes so that every profiled launch runs the policy the bench line was measured with")
    ap.add_argument("--placement-trials", type=int, default=10,
                    help="autotune: observation buffers tried (the fastest placement is kept, the others freed; 1 = keep "
                         "the first allocation)")
    ap.add_argument("--obs-dtype", choices=["int32", "u8"], default="int32",
                    help="grid workloads with a shared layout: u8 = the narrow observation (one byte per cell, lmaze_step_u8), a "
                         "separate workload with its own algorithmic bytes 37 + G*G; int32 is the metric's mode")
    ap.add_argument("--one-launch", action="store_true",
                    help="c2: the K timed steps as ONE lmaze_rollout call (a wave keeps its envs in registers across the steps; "
                         "launch-bound sizes) instead of K step launches")
    ap.add_argument("--graph", action="store_true",
                    help="capture the K timed launches into one hipGraph and time its replay (launch-bound sizes)")
    ap.add_argument("--action-rows", type=int, default=None,
                    help="rows of the pre-generated action tensor int32[rows, N], cycled (row t %% rows at step t). "
                         "Default, whatever --steps is: enough rows for 320 MiB (80 rows at 1M envs), more than the "
                         "256 MiB Infinity Cache holds, so every step's row comes from HBM; c2: the 256 rows of SURVEY "
                         "8(d) C2.  32 rows of 1M envs are 134 MB, stay in the cache and flatter the step kernel.")
    return ap.parse_args(argv)


def default_action_rows(workload, N):
    if workload == "c2":
        return 256                                   # actions int32[T=256, N]
    return max(2, -(-(320 << 20) // (4 * N)))        # > the 256 MiB Infinity Cache, whatever --steps is


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process only starts the N ranks (no torch, no HIP in it)
        sys.exit(spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    import numpy as np
    import torch

    foveal = args.workload in FOVEAL_ACTIONS
    hier = args.workload == "v5"
    preset = {"c3": (1 << 20, 11, False), "c2": (65536, 8, False), "c5": (1 << 20, 32, True),
              "v1": (1 << 20, 14, False), "v2": (1 << 20, 18, False), "v4": (1 << 20, 18, False),
              "v5": (1 << 20, 18, False)}[args.workload]
    if foveal and (args.graph or args.per_env_layouts or args.grid is not None):
        raise SystemExit("--graph / --per-env-layouts / --grid do not apply to the foveal workloads")
    if hier and args.auto_reset:
        raise SystemExit("--workload v5 is the two-level step: the reset is always fused in")
    args.envs = args.envs if args.envs is not None else preset[0]
    args.grid = args.grid if args.grid is not None else preset[1]
    args.per_env_layouts = args.per_env_layouts or preset[2]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d inside a process group of %d ranks: start it as `python bench.py --gpus N` "
                         "or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernels are the only step path")
    # LMAZE_BENCH_BACKEND=gloo is a rehearsal switch: several ranks share the visible GPU(s) and the barrier /
    # MAX go over gloo, to exercise the multi-rank code path on a one-GPU box.  Never used for reported numbers.
    backend = os.environ.get("LMAZE_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend != "nccl":
        local_rank %= max(1, ndev)
    elif local_rank >= ndev:
        raise SystemExit("bench.py --gpus %d: rank %d has no device (this node shows %d); one process per GPU -- "
                         "LMAZE_BENCH_BACKEND=gloo rehearses the multi-rank path on fewer GPUs" % (args.gpus, rank, ndev))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # one of several ranks (self-launched or torch.distributed.run)
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
                dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm; barrier + MAX + one gather only
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
    tuned, layout, goals = None, None, None
    R = args.action_rows if args.action_rows else default_action_rows(args.workload, N)
    gen = torch.Generator(device=dev).manual_seed(1 + rank)      # torch's device generator is Philox
    if foveal:
        variant = args.workload
        env = pkg.LmazeFovealVecEnv(N, variant=variant, device=dev, seed=1, env_base=env_base)
        G = env.grid
        if args.launch_hint is not None:
            env.params.launch_hint = args.launch_hint
        if variant == "v1":        # the two-level loop's upper half: a foveal goal per env (lmaze_env_v1.py:104-110)
            env.set_foveal_goal(torch.randint(0, 5, (N, 2), dtype=torch.int32, device=dev, generator=gen))
        workload = ("%d x lmaze-%s per GPU (%dx%d layouts, 5x5 window, float32 [%d,5,5] observation%s)"
                    % (N, variant, G, G, env.channels,
                       ", float32 visit map" if variant == "v4" else
                       " + float32 [4,5,5] local observation, float32 visit map; two-level loop: reset on globalDone, "
                       "plannerStep(uniform goal) on localDone, step(uniform action) -- one launch per env-step" if hier else ""))
        actions = torch.randint(0, FOVEAL_ACTIONS[variant], (R, N), dtype=torch.int32, device=dev, generator=gen)
        row_ptr = [actions[r].data_ptr() for r in range(R)]
        if hier:
            goals = torch.randint(0, 25, (R, N), dtype=torch.int32, device=dev, generator=gen)
            goal_ptr = [goals[r].data_ptr() for r in range(R)]
            env.foveal_done.fill_(True)        # every env starts with a plannerStep, as the two-level loop does

            def run(k0, k, captured=False):
                for t in range(k0, k0 + k):
                    env.hier_step_raw(row_ptr[t % R], goal_ptr[t % R])
        else:
            def run(k0, k, captured=False):
                for t in range(k0, k0 + k):
                    env.step_raw(row_ptr[t % R], auto_reset=args.auto_reset)
        # warm the device as LmazeVecEnv.autotune() does for the grid workloads (cold clocks)
        if args.launch_hint is not None or args.no_autotune:
            warm_device(pkg, env.obs.numel() * 4, dev)     # a different kernel: see the grid branch below
        else:
            with torch.cuda.device(dev):
                run(0, 150)
        if args.launch_hint is None and not args.no_autotune:
            # untimed, state restored: (envs per workgroup, workgroups per CU) for this device, on the tensors the timed
            # steps read -- as LmazeVecEnv.autotune() does for the grid workloads
            tuned = env.autotune(actions, goals=goals, auto_reset=args.auto_reset, placement_trials=args.placement_trials)
    else:
        if args.workload == "c2":
            layout = pkg.layouts.to_codes(pkg.layouts.GRID_8_BORDERED)   # lmaze_env.py:28-35 literal, bordered
        else:
            layout = pkg.layouts.to_codes(pkg.layouts.open_room(G, (G // 2, G // 2)))
        if args.per_env_layouts:
            # border 'W', interior walls i.i.d. p = 0.25 (Philox seed 7), 'X' on a uniformly chosen free
            # cell, the ball on another (reset(): uniform over the cells that are neither 'W' nor 'X')
            lay = pkg.layouts.random_walled(N, G, dev, p_wall=0.25, seed=7 + rank)
            env = pkg.LmazeVecEnv(N, variant="v0", per_env_layouts=lay, device=dev, seed=1, env_base=env_base)
            workload = ("%d x %dx%d mazes per GPU, v0 rules, per-env random layouts (border 'W', interior walls i.i.d. p 0.25, "
                        "Philox seed 7+rank, 'X' on a uniformly chosen free cell, ball on another), compact int32 obs" % (N, G, G))
        else:
            env = pkg.LmazeVecEnv(N, variant="v0", layout=layout, device=dev, seed=1, env_base=env_base, obs_dtype=args.obs_dtype)
            workload = "%d x %dx%d mazes per GPU, v0 rules, shared layout: %s, compact %s obs" % (
                N, G, G, "the 8x8 literal of lmaze_env.py:28-35 with a 'W' border" if args.workload == "c2" and G == 8
                else "open room with a 'W' border, 'S' at (1,1), 'X' at (%d,%d)" % (G // 2, G // 2),
                "uint8 (NOT the metric's mode: 37 + G*G bytes per env-step)" if args.obs_dtype == "u8" else "int32")

        actions = torch.randint(0, 4, (R, N), dtype=torch.int32, device=dev, generator=gen)
        row_ptr = [actions[r].data_ptr() for r in range(R)]
        if args.launch_hint is not None:
            env.params.launch_hint = args.launch_hint
        if args.launch_hint is None and not args.no_autotune:
            # untimed: picks (workgroups per CU, chunks per workgroup) for this shape and device, on the very
            # action tensor the timed steps read (cache-resident or not decides the ranking)
            tuned = env.autotune(auto_reset=args.auto_reset, actions=actions, placement_trials=args.placement_trials)
        else:
            # no autotune, so nothing has warmed the device yet: the first ~100 launches of a process run 10-40 % slow
            # (clocks still ramping) and --warmup alone would leave them in the timed region.  Warm
            # with a DIFFERENT kernel (the fill probe, same bytes per launch) so that under `rocprofv3 --stats` every
            # launch of the step kernel, warm-up included, is a warm one and the average is the timed policy's.
            warm_device(pkg, env.obs.numel() * 4, dev)

        def run(k0, k, captured=False):
            for t in range(k0, k0 + k):
                # under capture the reset epoch is a device word handed from launch to launch (slot = launch index)
                env.step_raw(row_ptr[t % R], auto_reset=args.auto_reset,
                             epoch_slot=(t - k0) if (captured and args.auto_reset) else None)

    snap = None
    with torch.cuda.device(dev):
        run(0, args.warmup)
        torch.cuda.synchronize()
        if hier or args.workload == "v4":   # the event counts of the timed steps come from a replay of exactly these steps (below)
            snap = env.snapshot()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        roll_actions = None
        if args.one_launch:
            if foveal or args.graph:
                raise SystemExit("--one-launch is lmaze_rollout of the grid workloads, without --graph")
            roll_actions = actions[(torch.arange(args.warmup, args.warmup + args.steps, device=dev) % R)].contiguous()
            env.rollout(roll_actions[:8].contiguous(), auto_reset=args.auto_reset)      # first call of the rollout kernel, untimed
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
        elif args.one_launch:
            # rows warmup .. warmup + steps - 1 of the ring, gathered into one contiguous int32[K, N] tensor beforehand
            env.rollout(roll_actions, auto_reset=args.auto_reset)
        else:
            run(args.warmup, args.steps)
        ev1.record()
        torch.cuda.synchronize()
        my_elapsed = time.perf_counter() - t0
        elapsed, per_rank = my_elapsed, [my_elapsed]
        if dist is not None:
            dist.barrier()
            elapsed = pkg.max_over_ranks(my_elapsed, device=dev)
            per_rank = pkg.gather_over_ranks(my_elapsed, device=dev)
    # per-launch GPU time from HIP events recorded on the launch stream
    kern_ms = float(ev0.elapsed_time(ev1) / args.steps)   # ms per launch, launch gaps included
    # every rank's own figures (a straggler in a multi-GPU run should explain itself): event time per launch, launch
    # policy in force, where its observation buffer ended up
    rank_kern_ms = pkg.gather_over_ranks(kern_ms, device=dev) if dist is not None else [kern_ms]
    rank_hint = pkg.gather_over_ranks(float(int(env.params.launch_hint)), device=dev) if dist is not None else [float(int(env.params.launch_hint))]
    rank_first_ms = None
    pl = getattr(env, "placement", None)
    if pl:
        rank_first_ms = pkg.gather_over_ranks(pl["trials_ms"][0], device=dev) if dist is not None else [pl["trials_ms"][0]]

    # sanity: the run really stepped (every env advanced warmup+steps times)
    if hier:       # local episodes restart at plannerStep: nobody is past the local step limit
        top = int(env.step_count.max().item())
        assert 1 <= top <= env.params.step_limit, top
    elif foveal:      # like the reference, stepping goes on past `done` unless the reset is fused in
        top = int(env.step_count.max().item())
        assert top >= 1 and (not args.auto_reset or top <= env.params.step_limit + 1)
    elif not args.auto_reset:
        assert int(env.step_count.min().item()) == args.warmup + args.steps + (8 if args.one_launch else 0)
    else:  # episodes restart: nobody is past the step limit, and everybody moved
        assert 1 <= int(env.step_count.min().item()) and int(env.step_count.max().item()) <= env.step_limit

    v5_events = None
    if hier and rank == 0:
        # replay the timed steps from the snapshot (same actions, goals and epochs => the same trajectory) and count
        # the events the algorithmic bytes depend on; untimed
        env.restore(snap)
        cnt = torch.zeros(5, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            for t in range(args.warmup, args.warmup + args.steps):
                fresh = env.done.clone()
                plan = fresh | env.foveal_done
                env.hier_step_raw(row_ptr[t % R], goal_ptr[t % R])
                upd = env.foveal_done
                cnt += torch.stack([plan.sum(), fresh.sum(), (upd & ~fresh).sum(), (~upd & ~fresh).sum(), upd.sum()])
        v5_events = dict(zip(("planner_steps", "resets", "visit_updates", "window_gathers", "local_dones"), cnt.tolist()))
    v4_events = None
    if args.workload == "v4" and rank == 0:
        # the same for v4: cells the "previous" window shows beside the current one (25 - overlap of the two 5x5 windows,
        # from the ball before and after each step) and, with --auto-reset, the resets
        env.restore(snap)
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            for t in range(args.warmup, args.warmup + args.steps):
                before, fresh = env.ball_xy.clone(), env.done.clone()
                env.step_raw(row_ptr[t % R], auto_reset=args.auto_reset)
                d = (env.ball_xy - before).abs().clamp(max=5)
                only = 25 - (5 - d[:, 0]) * (5 - d[:, 1])
                if args.auto_reset:
                    only = torch.where(fresh, torch.zeros_like(only), only)     # a reset shows the same window twice
                cnt += torch.stack([only.sum(), fresh.sum() if args.auto_reset else fresh.sum() * 0])
        v4_events = dict(zip(("previous_only_cells", "resets"), cnt.tolist()))

    ceiling = measured_ceiling(pkg, env.obs.numel() * 4, dev) if rank == 0 else None

    if rank == 0:
        if hier:
            ev = v5_events
            B = V5_BASE + (V5_PLAN * ev["planner_steps"] + V5_RESET * ev["resets"] + V5_UPDATE * ev["local_dones"]) / float(N * args.steps)
        elif args.workload == "v4":
            B = FOVEAL_BYTES["v4"] + (V4_PREV_ONLY_CELL * v4_events["previous_only_cells"]
                                      + V4_RESET * v4_events["resets"]) / float(N * args.steps)
        else:
            B = FOVEAL_BYTES[args.workload] if foveal else bytes_per_env_step(G, args.per_env_layouts)
            if args.obs_dtype == "u8" and not foveal:
                B = 37 + G * G
        total_steps = world * N * args.steps
        value = total_steps / elapsed
        # ONE clock: achieved / frac follow from the line's own ms_per_step (host wall time around the timed launches,
        # max over ranks; every GPU moves N * B bytes per step); the HIP-event figure of rank 0 sits beside it
        achieved = N * B / (elapsed / args.steps) / 1e9
        achieved_events = N * B / (kern_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = args.workload if foveal else "g%d_%s" % (G, "perenv" if args.per_env_layouts else "shared")
                rec = tj.get(key, {})
                # the PMC passes profiled one launch shape: only quote them for that shape
                alg = rec.get("algorithmic_bytes_per_launch")
                if alg is not None and abs(alg - N * B) <= 0.02 * N * B:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = ("profiles/traffic.json['%s']: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of round %s "
                                      "on this workload (%s), NOT measured by this run"
                                      % (key, rec.get("round"), rec.get("files", "profiles/r%02d/" % int(rec.get("round", 1)))))
            except Exception:
                traffic, traffic_source = None, None
        # the kernel and policy the launcher actually picks for this shape and hint (lmaze_describe_*: the launcher's own
        # decision code, nothing queued)
        abi = importlib.import_module(pkg.__name__ + "._abi")
        if foveal:
            kernel = "lmaze::" + abi.describe_foveal_step(env.params, N, auto_reset=bool(args.auto_reset) or hier)
            perenv_kernel = None
        else:
            kernel = "lmaze::" + abi.describe_step(env.params, N, auto_reset=bool(args.auto_reset),
                                                   with_obs="u8" if args.obs_dtype == "u8" else True)
            if args.one_launch:
                kernel += (" -- timed as ONE lmaze_rollout call of %d steps (shared layouts: rollout_shared_wave8_kernel for on-die "
                           "8x8 batches, rollout_shared_kernel otherwise; per-env layouts: rollout_perenv_kernel while the planes "
                           "stay on-die, else that many launches inside the call)" % args.steps)
            perenv_kernel = None
            if args.per_env_layouts:
                # BASELINE config 5 names an LDS-tiled maze per workgroup; at G*G a multiple of 256 the register-tiled
                # one-wave-per-env kernel is used instead because it measured faster (0.873 vs 0.96 ms)
                perenv_kernel = ("wave/register-tiled (one wave per env, layout in registers; measured faster than the LDS-tiled "
                                 "kernel at this G)" if "perenv_wave" in kernel else "LDS-tiled (a workgroup tiles ~8 KiB of layouts)")
        out = {
            "metric": "env steps/sec (whole node), 1M parallel 11x11 mazes at 1/2/4/8 MI355X" if args.workload == "c3" and N == (1 << 20) and args.obs_dtype == "int32"
                      else "env steps/sec (whole node); workload '%s', NOT the configuration BASELINE.json's metric is quoted on" % args.workload,
            "workload_id": args.workload,
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if foveal else ("u8" if args.obs_dtype == "u8" else "int32"), "data": "synthetic",
            "per_rank_ms_per_step": {"min": min(per_rank) / args.steps * 1e3, "max": max(per_rank) / args.steps * 1e3,
                                     "all": [round(x / args.steps * 1e3, 6) for x in per_rank]},
            "config": {"workload": workload, "envs_per_gpu": N, "grid": G, "global_envs": world * N,
                       "parallelism": "independent env shards, no collective on the step path",
                       "world_size_seen": int(dist.get_world_size()) if dist is not None else 1,
                       "launcher": ("self (bench.py spawned its ranks)" if os.environ.get("LMAZE_BENCH_SELF_LAUNCHED")
                                    else "external (torch.distributed.run)") if dist is not None else "single process",
                       "actions": "uniform{0..%d} int32[%d,N] (%d MiB) on the device, row t %% rows at step t, torch Philox seed 1+rank"
                                  % (FOVEAL_ACTIONS[args.workload] - 1 if foveal else 3, R, (R * N * 4) >> 20),
                       "auto_reset": bool(args.auto_reset) or hier, "hip_graph": bool(args.graph),
                       "one_launch_rollout": bool(args.one_launch),
                       "collective_backend": ("rccl" if backend == "nccl" else backend + " (REHEARSAL, ranks share a GPU)")
                       if dist is not None else None,
                       "launch_hint": int(env.params.launch_hint),
                       "perenv_kernel": perenv_kernel,
                       "obs_placement": getattr(env, "placement", None),
                       "autotune_ms": {("x".join(str(int(x)) for x in k) if isinstance(k, tuple) else ("0x%02x" % k if foveal else str(k))): round(v, 5)
                                       for k, v in (tuned or {}).items()}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel, "bytes_per_env_step": B, "kernel_ms_avg": kern_ms,
                         "achieved_events": achieved_events, "frac_events": achieved_events / HBM_PEAK_GBS,
                         # what a caller who never tries placements of the observation buffer gets (library default
                         # placement_trials = 0): the first allocation under the tuned policy, HIP events, from autotune()
                         "frac_first_allocation": (N * B / (pl["first_ms_tuned"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                   if pl and pl.get("first_ms_tuned") else None),
                         "frac_kept_allocation": (N * B / (pl["kept_ms_tuned"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                  if pl and pl.get("kept_ms_tuned") else None),
                         # the untuned library: first allocation AND launch_hint 0 (the per-shape default of the policy table)
                         "frac_untuned_library": (N * B / (pl["first_ms_default"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                  if pl and pl.get("first_ms_default") else None),
                         "note": ("on-die: the planes of this batch never leave L2 / the Infinity Cache, so the HBM fraction is "
                                  "nominal (it can exceed 1)" if args.one_launch else
                                  "frac is priced on algorithmic bytes; traffic is what the PMC counters saw for this launch shape"),
                         "clock": "achieved / frac: ms_per_step (perf_counter around the timed launches, max over ranks); "
                                  "achieved_events / frac_events / kernel_ms_avg: HIP events on rank 0's launch stream",
                         "measured_ceiling": ceiling},
        }
        out["per_rank"] = {
            "ms_per_step": [round(x / args.steps * 1e3, 6) for x in per_rank],
            "kernel_ms_avg": [round(x, 6) for x in rank_kern_ms],
            "launch_hint": [int(x) for x in rank_hint],
            "roofline_frac": [round(N * B / (x / args.steps) / 1e9 / HBM_PEAK_GBS, 4) for x in per_rank],
            "roofline_frac_events": [round(N * B / (x * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for x in rank_kern_ms],
            "note": "one entry per rank, rank order; roofline.frac above is the slowest rank's (ms_per_step = max over ranks)"}
        if hier:
            out["config"]["v5_events_in_timed_steps"] = v5_events
            out["config"]["local_done_rate"] = v5_events["local_dones"] / float(N * args.steps)
        if v4_events is not None:
            out["config"]["v4_events_in_timed_steps"] = v4_events
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N=1 only
            out["cpu_baseline"] = (cpu_baseline_foveal(args.workload, args.cpu_baseline_seconds) if foveal
                                   else cpu_baseline(args.workload, G, layout, args.cpu_baseline_seconds))
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
