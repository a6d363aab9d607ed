"""LmazeFovealVecEnv: N envs of the foveal variants (v1, v2, v4) as struct-of-arrays torch
tensors in HBM, stepped by one HIP kernel per step() through lmaze_foveal_* (include/lmaze.h).

Host side only (buffers, marshalling, stream); the transition, the visit-map update and the
5x5 window render run in liblmaze_hip.so.  Reference: gym_lmaze/envs/lmaze_env_v1.py,
lmaze_env_v2.py, lmaze_env_v4.py.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi
from . import layouts as L
from .vec_env import _align, resolve_device

FOVEAL_VARIANTS = {
    # lmaze_env_v1.py:21-36
    "v1": dict(id=_abi.VARIANT_V1, layouts=(L.V1_GRID_14,), channels=4, expansion=7, step_limit=200,
               foveal_step_limit=10, rewards=(-1.0, 0.01, 1.0), n_actions=4),
    # lmaze_env_v2.py:26-49
    "v2": dict(id=_abi.VARIANT_V2, layouts=L.FOVEAL_GRIDS_18, channels=5, expansion=7, step_limit=50,
               foveal_step_limit=0, rewards=(-1.0, -0.01, 100.0), n_actions=25),
    # lmaze_env_v4.py:23-48
    "v4": dict(id=_abi.VARIANT_V4, layouts=L.FOVEAL_GRIDS_18, channels=7, expansion=7, step_limit=50,
               foveal_step_limit=0, rewards=(-1.0, -0.01, 100.0), n_actions=25),
    # lmaze_env_v5.py:21-49: local step limit 10 (>=), foveal step limit 50 (>=); 4 local actions, 25 planner goals
    "v5": dict(id=_abi.VARIANT_V5, layouts=L.FOVEAL_GRIDS_18, channels=7, expansion=7, step_limit=10,
               foveal_step_limit=50, rewards=(-1.0, -0.01, 100.0), n_actions=4),
    "v6": dict(id=_abi.VARIANT_V6, layouts=L.FOVEAL_GRIDS_18, channels=7, expansion=7, step_limit=10,
               foveal_step_limit=50, rewards=(-1.0, -0.01, 100.0), n_actions=4),
}


class LmazeFovealVecEnv(object):
    """N foveal mazes.  obs is float32[N,C,5,5] -- the reference's retState before its x7 loop;
    expanded() gives the (C,35,35) reference layout."""

    def __init__(self, num_envs, variant="v2", layouts=None, device=None, seed=0, env_base=0, reset=True):
        if variant not in FOVEAL_VARIANTS:
            raise ValueError("unknown foveal variant %r (have %s)" % (variant, sorted(FOVEAL_VARIANTS)))
        spec = FOVEAL_VARIANTS[variant]
        self.variant = variant
        self.num_envs = N = int(num_envs)
        if N < 1:
            raise ValueError("num_envs must be >= 1")
        self.device = resolve_device(device)
        self.channels = spec["channels"]
        self.expansion = spec["expansion"]
        self.seed, self.env_base, self._epoch = int(seed), int(env_base), 0
        self.tuned_policy = None     # launch_hint chosen by autotune()
        self.placement = None        # autotune(placement_trials=K): where the observation buffer ended up
        tabs = [L.to_codes(t) for t in (layouts if layouts is not None else spec["layouts"])]
        G = tabs[0].shape[0]
        pad = 2 if variant == "v1" else 2   # the 5x5 window must stay inside the array
        for t in tabs:
            if t.shape != (G, G):
                raise ValueError("all layouts must share one square shape")
            L.validate(t, need_goal_marker=(variant == "v1"))
            W = ord("W")
            if not ((t[:pad] == W).all() and (t[-pad:] == W).all() and (t[:, :pad] == W).all() and (t[:, -pad:] == W).all()):
                raise ValueError("foveal layouts need a %d-cell 'W' padding (lmaze_env_v1.py:40-53)" % pad)
        if variant == "v1" and len(tabs) != 1:
            raise ValueError("v1 has a single layout")
        self.grid = G
        self.n_layouts = len(tabs)
        self.layouts = torch.from_numpy(np.stack(tabs).copy()).to(self.device)

        sizes = [("ball_xy", 8 * N), ("goal_xy", 8 * N), ("fgoal_xy", 8 * N), ("layout_id", 4 * N),
                 ("step_count", 4 * N), ("foveal_step_count", 4 * N), ("reward", 4 * N), ("foveal_reward", 4 * N),
                 ("done", N), ("foveal_done", N), ("ball1_xy", 8 * N), ("fovea_xy", 16 * N), ("last_xy", 8 * N),
                 ("foveal_goal", 4 * N)]
        offs, total = {}, 0
        for name, sz in sizes:
            offs[name] = total
            total += _align(sz)
        self._state = torch.zeros(total, dtype=torch.uint8, device=self.device)
        # device-resident epoch pair of captured auto-reset / two-level steps; allocated here, never under capture
        # (an allocation inside a capture becomes a memset node that every replay would re-run)
        self._epoch_words = torch.zeros(2, dtype=torch.int64, device=self.device)

        def view(name, nbytes, dtype, shape):
            return self._state[offs[name]:offs[name] + nbytes].view(dtype).view(shape)

        self.ball_xy = view("ball_xy", 8 * N, torch.int32, (N, 2))
        self.goal_xy = view("goal_xy", 8 * N, torch.int32, (N, 2))
        self.fgoal_xy = view("fgoal_xy", 8 * N, torch.int32, (N, 2))
        self.layout_id = view("layout_id", 4 * N, torch.int32, (N,))
        self.step_count = view("step_count", 4 * N, torch.int32, (N,))
        self.foveal_step_count = view("foveal_step_count", 4 * N, torch.int32, (N,))
        self.reward = view("reward", 4 * N, torch.float32, (N,))
        self.foveal_reward = view("foveal_reward", 4 * N, torch.float32, (N,))
        self._done_u8 = view("done", N, torch.uint8, (N,))
        self._fdone_u8 = view("foveal_done", N, torch.uint8, (N,))
        self.done = self._done_u8.view(torch.bool)
        self.foveal_done = self._fdone_u8.view(torch.bool)
        self.ball1_xy = view("ball1_xy", 8 * N, torch.int32, (N, 2))
        self.fovea_xy = view("fovea_xy", 16 * N, torch.int32, (N, 4))
        self.last_xy = view("last_xy", 8 * N, torch.int32, (N, 2))
        self.foveal_goal = view("foveal_goal", 4 * N, torch.int32, (N,))
        self._two_level = variant in ("v5", "v6")
        # the visit map in the library's clock-relative 4x4 tiles (include/lmaze.h "The visit map"): opaque here; the
        # reference's float[N,G,G] is the `visit` property (lmaze_foveal_materialise_visit)
        self._has_visit = variant in ("v4", "v5", "v6")
        self._visit_tiles = self._visit_clock = None
        if self._has_visit:
            nbytes = int(_abi.lib.lmaze_foveal_visit_bytes(G, N))
            self._visit_tiles = torch.zeros(nbytes // 4, dtype=torch.float32, device=self.device)
            self._visit_clock = torch.zeros(N, dtype=torch.int32, device=self.device)
            assert self._visit_tiles.data_ptr() % 64 == 0
        self.obs_local = (torch.zeros((N, 4, _abi.FOVEA, _abi.FOVEA), dtype=torch.float32, device=self.device)
                          if self._two_level else None)
        self._expanded_local = None
        self.obs = torch.zeros((N, self.channels, _abi.FOVEA, _abi.FOVEA), dtype=torch.float32, device=self.device)
        self._expanded = None
        self._names = sizes

        r = spec["rewards"]
        self.params = _abi.LmazeFovealParams(spec["id"], G, self.n_layouts, spec["step_limit"],
                                             spec["foveal_step_limit"], r[0], r[1], r[2], 0)
        self._pp = C.byref(self.params)
        self.bufs = _abi.LmazeFovealBuffers(
            self.ball_xy.data_ptr(), self.goal_xy.data_ptr(), self.fgoal_xy.data_ptr(), self.layout_id.data_ptr(),
            self.step_count.data_ptr(), self.foveal_step_count.data_ptr(), self.reward.data_ptr(),
            self.foveal_reward.data_ptr(), self._done_u8.data_ptr(), self._fdone_u8.data_ptr(),
            self._visit_tiles.data_ptr() if self._has_visit else None, self.obs.data_ptr(),
            self.ball1_xy.data_ptr(), self.fovea_xy.data_ptr(), self.last_xy.data_ptr(), self.foveal_goal.data_ptr(),
            self.obs_local.data_ptr() if self.obs_local is not None else None,
            self._visit_clock.data_ptr() if self._has_visit else None)
        self._pb = C.byref(self.bufs)
        self._p_layouts = self.layouts.data_ptr()
        if variant == "v1":
            # goal = the 'X' cell (lmaze_env_v1.py:86-88); kept for the drop-in attributes
            c = int(np.flatnonzero(tabs[0].reshape(-1) == ord("X"))[0])
            self.goal_xy.copy_(torch.tensor([[c // G, c % G]], dtype=torch.int32).expand(N, 2))
        if reset:
            self.reset()

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _guard(self):
        return torch.cuda.device(self.device)

    def _as_i32(self, x, numel):
        if isinstance(x, torch.Tensor):
            t = x.to(device=self.device, dtype=torch.int32)
        else:
            t = torch.as_tensor(np.asarray(x, dtype=np.int64).astype(np.int32), device=self.device)
        t = t.reshape(-1).contiguous()
        if t.numel() != numel:
            raise ValueError("expected %d values, got %d" % (numel, t.numel()))
        return t

    def _mask_ptr(self, mask):
        if mask is None:
            return None, None
        m = mask if isinstance(mask, torch.Tensor) else torch.as_tensor(np.asarray(mask), device=self.device)
        m = m.to(device=self.device)
        m = (m.view(torch.uint8) if m.dtype == torch.bool else (m != 0).to(torch.uint8)).contiguous()
        if m.numel() != self.num_envs:
            raise ValueError("mask must have %d entries" % self.num_envs)
        return m, m.data_ptr()

    # ------------------------------------------------------------------ the visit map (v4, v5, v6)
    @property
    def visit(self):
        """The reference's state[2]: float32[N,G,G] true values, materialised from the clock-relative tiles (a NEW
        tensor on every access; off the step path).  None for v1 / v2."""
        if not self._has_visit:
            return None
        out = torch.empty((self.num_envs, self.grid, self.grid), dtype=torch.float32, device=self.device)
        with self._guard():
            rc = _abi.lib.lmaze_foveal_materialise_visit(self._pp, self._pb, out.data_ptr(), self.num_envs, self._stream())
        _abi.check("lmaze_foveal_materialise_visit", rc)
        return out

    def load_visit(self, values):
        """Take float32[N,G,G] true values (e.g. the reference's own state[2]) as the visit map."""
        t = values if isinstance(values, torch.Tensor) else torch.as_tensor(np.asarray(values, dtype=np.float32))
        t = t.to(device=self.device, dtype=torch.float32).reshape(self.num_envs, self.grid, self.grid).contiguous()
        with self._guard():
            rc = _abi.lib.lmaze_foveal_load_visit(self._pp, self._pb, t.data_ptr(), self.num_envs, self._stream())
        _abi.check("lmaze_foveal_load_visit", rc)

    def snapshot(self):
        """Everything a step reads and writes except the observations: per-env scalars, visit tiles + clocks, epoch."""
        return (self._state.clone(), self._visit_tiles.clone() if self._has_visit else None,
                self._visit_clock.clone() if self._has_visit else None, self._epoch)

    def restore(self, snap):
        self._state.copy_(snap[0])
        if self._has_visit:
            self._visit_tiles.copy_(snap[1])
            self._visit_clock.copy_(snap[2])
        self._epoch = snap[3]

    # ------------------------------------------------------------------ the path
    def begin_replay(self, n_launches):
        """Before replaying a captured sequence of n_launches auto-reset steps (step(..., epoch_slot=t)):
        hand the host's epoch count to the device word launch 0 reads and reserve n_launches epochs."""
        self._epoch_words[0:1].fill_(self._epoch)
        self._epoch += int(n_launches)

    def step(self, actions, auto_reset=False, epoch_slot=None):
        """v1: ids 0..3 (else no move); v2/v4: 0..24 = 5*row+col of the target cell in the window.
        Returns (obs, reward, done, actions); v1's second stream is in foveal_reward / foveal_done.
        auto_reset=True (v1, v2, v4) first resets the envs whose done flag is still set from the previous
        step, fused into the same kernel: bit-identical to reset(mask=done) + step(actions).
        epoch_slot=t (under hipGraph capture, t = index of the launch in the captured sequence) keeps the
        reset epoch on the device so that replays draw fresh placements; call begin_replay(T) before each."""
        a = self._as_i32(actions, self.num_envs)
        with self._guard():
            if auto_reset:
                if self._two_level:
                    raise ValueError("auto_reset is not defined for v5/v6 (episodes restart through planner_step)")
                if epoch_slot is None:
                    epoch, e_in, e_out = self._epoch, None, None
                    self._epoch += 1
                else:
                    base, t = self._epoch_words.data_ptr(), int(epoch_slot)
                    epoch, e_in, e_out = 0, base + 8 * (t & 1), base + 8 * ((t + 1) & 1)
                rc = _abi.lib.lmaze_foveal_step_autoreset(self._pp, self._p_layouts, a.data_ptr(), self._pb,
                                                          self.num_envs, self.seed & (2 ** 64 - 1), epoch,
                                                          self.env_base, e_in, e_out, self._stream())
            else:
                rc = _abi.lib.lmaze_foveal_step(self._pp, self._p_layouts, a.data_ptr(), self._pb, self.num_envs,
                                                self._stream())
        _abi.check("lmaze_foveal_step", rc)
        return self.obs, self.reward, self.done, actions

    def step_raw(self, action_ptr, auto_reset=False, epoch_slot=None):
        """step() on a raw device pointer to int32[N] actions (rollouts over a pre-generated [T,N] tensor)."""
        with self._guard():
            if auto_reset:
                if self._two_level:
                    raise ValueError("auto_reset is not defined for v5/v6 (use hier_step)")
                if epoch_slot is None:
                    epoch, e_in, e_out = self._epoch, None, None
                    self._epoch += 1
                else:
                    base, t = self._epoch_words.data_ptr(), int(epoch_slot)
                    epoch, e_in, e_out = 0, base + 8 * (t & 1), base + 8 * ((t + 1) & 1)
                rc = _abi.lib.lmaze_foveal_step_autoreset(self._pp, self._p_layouts, action_ptr, self._pb, self.num_envs,
                                                          self.seed & (2 ** 64 - 1), epoch, self.env_base,
                                                          e_in, e_out, self._stream())
            else:
                rc = _abi.lib.lmaze_foveal_step(self._pp, self._p_layouts, action_ptr, self._pb, self.num_envs, self._stream())
        _abi.check("lmaze_foveal_step", rc)

    def hier_step(self, actions, goals, epoch_slot=None):
        """v5/v6: the two-level loop around step() as ONE launch (lmaze_v5_hier_step) -- reset() for the envs whose
        globalDone (`done`) is set on entry, plannerStep(goals[i]) for those whose localDone (`foveal_done`) is set
        or that were just reset, then step(actions[i]) for everybody.  Bit-identical to reset(mask=done),
        planner_step(goals, mask=done | foveal_done), step(actions).  Returns (obs, obs_local, reward,
        foveal_reward, done, foveal_done) -- the tensors of the reference's 8-tuple (lmaze_env_v5.py:285-292)."""
        a = self._as_i32(actions, self.num_envs)
        g = self._as_i32(goals, self.num_envs)
        self.hier_step_raw(a.data_ptr(), g.data_ptr(), epoch_slot)
        return self.obs, self.obs_local, self.reward, self.foveal_reward, self.done, self.foveal_done

    def hier_step_raw(self, action_ptr, goal_ptr, epoch_slot=None):
        if not self._two_level:
            raise ValueError("hier_step is the two-level loop of v5/v6")
        with self._guard():
            if epoch_slot is None:
                epoch, e_in, e_out = self._epoch, None, None
                self._epoch += 1
            else:
                base, t = self._epoch_words.data_ptr(), int(epoch_slot)
                epoch, e_in, e_out = 0, base + 8 * (t & 1), base + 8 * ((t + 1) & 1)
            rc = _abi.lib.lmaze_v5_hier_step(self._pp, self._p_layouts, action_ptr, goal_ptr, self._pb, self.num_envs,
                                             self.seed & (2 ** 64 - 1), epoch, self.env_base, e_in, e_out, self._stream())
        _abi.check("lmaze_v5_hier_step", rc)

    def rollout(self, actions, goals=None, auto_reset=False, device_epoch=False):
        """T steps over device tensors int32[T,N], one kernel per step, no host sync: step(actions[t]) -- with the
        reset fused in when auto_reset -- or, for v5/v6 with `goals`, the two-level step hier_step(actions[t], goals[t]).
        device_epoch: keep the reset epoch on the device (what capture_rollout uses)."""
        hier = goals is not None
        for t in (actions, goals):
            if t is not None and not (isinstance(t, torch.Tensor) and t.dtype == torch.int32 and t.dim() == 2
                                      and t.shape[1] == self.num_envs and t.device == self.device and t.is_contiguous()):
                raise ValueError("rollout() wants contiguous int32[T,N] tensors on %s" % (self.device,))
        if hier and (not self._two_level or goals.shape[0] != actions.shape[0]):
            raise ValueError("planner goals: v5/v6 only, one row per step")
        if self._two_level and not hier:
            if auto_reset:
                raise ValueError("v5/v6 restart episodes through the two-level step: pass goals")
        base, stride = actions.data_ptr(), self.num_envs * 4
        for t in range(int(actions.shape[0])):
            slot = t if device_epoch else None
            if hier:
                self.hier_step_raw(base + t * stride, goals.data_ptr() + t * stride, slot)
            else:
                self.step_raw(base + t * stride, auto_reset=auto_reset, epoch_slot=slot if auto_reset else None)
        return self.obs, self.reward, self.done

    def capture_rollout(self, actions, goals=None, auto_reset=False):
        """rollout(actions, goals, auto_reset) captured into ONE hipGraph (a RolloutGraph; call .replay()): for
        launch-bound batch sizes.  With resets in it (auto_reset, or the two-level step) the epoch is a device word
        the launches hand on to each other, so every replay draws fresh placements -- exactly those the same steps
        launched eagerly would draw."""
        from .vec_env import RolloutGraph
        needs_epoch = bool(auto_reset) or goals is not None
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                self.rollout(actions, goals=goals, auto_reset=auto_reset, device_epoch=needs_epoch)
        torch.cuda.current_stream(self.device).wait_stream(side)
        self._captured = getattr(self, "_captured", 0) + 1
        return RolloutGraph(self, graph, int(actions.shape[0]), needs_epoch)

    # launch policies autotune() tries: LmazeFovealParams.launch_hint = ((chunks per workgroup - 1) << 8) |
    # (envs-per-workgroup code << 4) | workgroups per CU
    CANDIDATES = (0x00, 0x20, 0x26, 0x30, 0x34, 0x35, 0x36, 0x43, 0x40, 0x45, 0x120, 0x220, 0x130, 0x134, 0x140, 0x145)

    def autotune(self, actions, goals=None, auto_reset=False, steps=24, candidates=None, warm=100, rounds=3,
                 placement_trials=0):
        """Pick LmazeFovealParams.launch_hint by timing real steps with HIP events on the caller's own action tensor
        (int32[T,N] on the device, rows cycled; `goals` likewise for the v5/v6 two-level step); state, visit maps and
        the observations are snapshotted and restored, so results are unaffected.  The best (envs per workgroup, workgroups per CU)
        pair moves from device to device -- v2 at 1M envs: uncapped 96 us on one box and 102 on the next, 6 per CU 95
        and 95, 5 per CU 99 and 90 -- exactly as for the grid kernels (LmazeVecEnv.autotune).  The median of `rounds`
        interleaved passes counts and the library default (hint 0) is kept unless another hint beats it by more than
        1.5 %.  placement_trials=K (K > 1): first the step is timed (uncapped policy, the sensitive one) on K - 1 further
        allocations of the observation buffer and the fastest becomes `self.obs` -- possibly a NEW tensor; as for
        LmazeVecEnv.autotune, where the driver placed the write target is worth up to 8 % uncapped and 2-3 % at the tuned
        cap (tools/placement_pmc.py, LAB_NOTES.md R3.2).  Returns {hint: ms}."""
        N = self.num_envs
        for t in (actions, goals):
            if t is not None and not (isinstance(t, torch.Tensor) and t.dtype == torch.int32 and t.dim() == 2
                                      and t.shape[1] == N and t.device == self.device and t.is_contiguous()):
                raise ValueError("autotune() wants contiguous int32[T,N] tensors on %s" % (self.device,))
        if self._two_level and goals is None:
            raise ValueError("v5/v6: autotune() times the two-level step and needs planner goals")
        cands = list(candidates or self.CANDIDATES)
        if int(placement_trials) > 1 and getattr(self, "_captured", 0):
            raise RuntimeError("autotune(placement_trials > 1) would move the observation buffer under %d captured rollout(s): "
                               "tune before capture_rollout()" % self._captured)
        R = int(actions.shape[0])
        snap = self.snapshot()
        obs_snap = (self.obs.clone(), None if self.obs_local is None else self.obs_local.clone())
        k = [0]

        def run(n):
            for _ in range(n):
                r = k[0] % R
                if self._two_level:
                    self.hier_step_raw(actions[r].data_ptr(), goals[r % goals.shape[0]].data_ptr())
                else:
                    self.step_raw(actions[r].data_ptr(), auto_reset=auto_reset)
                k[0] += 1

        timings = {}
        with self._guard():
            run(int(warm))
            if int(placement_trials) > 1:
                bufs = [self.obs] + [torch.empty_like(self.obs) for _ in range(int(placement_trials) - 1)]
                self.params.launch_hint = 0x30
                ms_of = []
                for b in bufs:
                    self.bufs.obs = b.data_ptr()
                    run(3)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run(12)
                    e1.record()
                    e1.synchronize()
                    ms_of.append(e0.elapsed_time(e1) / 12)
                keep = min(range(len(bufs)), key=lambda i: ms_of[i])
                self.placement = {"trials_ms": [round(m, 5) for m in ms_of], "kept": keep}
                first_alloc = self.obs.view_as(self.obs) if keep != 0 else None    # keeps the first allocation alive (timed below)
                if keep != 0:
                    self.obs.set_(bufs[keep])      # the same tensor object on the winning allocation: held references stay valid
                self.bufs.obs = self.obs.data_ptr()
                self._expanded = None
                del bufs
            for _round in range(int(rounds)):          # interleaved passes; the median of a candidate's passes counts
                for h in cands:
                    self.params.launch_hint = int(h)
                    run(1)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run(steps)
                    e1.record()
                    e1.synchronize()
                    ms = e0.elapsed_time(e1) / steps
                    timings.setdefault(h, []).append(ms)
            if int(placement_trials) > 1:
                # what a caller who never tries placements gets: the FIRST allocation under the policy about to be chosen
                med = {h: sorted(v)[len(v) // 2] for h, v in timings.items()}
                b0 = min(med, key=med.get)
                if 0 in med and med[b0] > 0.985 * med[0]:
                    b0 = 0
                self.params.launch_hint = int(b0)
                kept_ptr = self.bufs.obs
                for name, ptr in (("kept_ms_tuned", kept_ptr),
                                  ("first_ms_tuned", first_alloc.data_ptr() if first_alloc is not None else kept_ptr)):
                    self.bufs.obs = ptr
                    run(3)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run(steps)
                    e1.record()
                    e1.synchronize()
                    self.placement[name] = round(e0.elapsed_time(e1) / steps, 5)
                # ... and under the library default policy (launch_hint 0): what an untuned LmazeFovealVecEnv runs
                self.params.launch_hint = 0
                self.bufs.obs = first_alloc.data_ptr() if first_alloc is not None else kept_ptr
                run(3)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(steps)
                e1.record()
                e1.synchronize()
                self.placement["first_ms_default"] = round(e0.elapsed_time(e1) / steps, 5)
                self.bufs.obs = kept_ptr
                first_alloc = None
            self.restore(snap)
            # the observations too: the tuning steps rendered into them, and with placement trials
            # self.obs may be another allocation by now -- the caller must see the frame of the restored state
            self.obs.copy_(obs_snap[0])
            if obs_snap[1] is not None:
                self.obs_local.copy_(obs_snap[1])
        timings = {h: sorted(v)[len(v) // 2] for h, v in timings.items()}
        best = min(timings, key=timings.get)
        if 0 in timings and timings[best] > 0.985 * timings[0]:
            best = 0
        self.params.launch_hint = int(best)
        self.tuned_policy = best
        return timings

    def reset(self, mask=None, place=True, seed=None):
        """Masked reset; place=False keeps the caller's ball/goal/layout_id (set_state)."""
        if seed is not None:
            self.seed, self._epoch = int(seed), 0
        m, m_ptr = self._mask_ptr(mask)
        with self._guard():
            rc = _abi.lib.lmaze_foveal_reset(self._pp, self._p_layouts, m_ptr, 1 if place else 0,
                                             self.seed & (2 ** 64 - 1), self._epoch, self.env_base, self._pb,
                                             self.num_envs, self._stream())
        _abi.check("lmaze_foveal_reset", rc)
        self._epoch += 1
        return self.obs

    def set_foveal_goal(self, ij, mask=None):
        """v1 setFovealGoal(i, j) (lmaze_env_v1.py:104-110); ij int[N,2]."""
        t = self._as_i32(ij, 2 * self.num_envs)
        m, m_ptr = self._mask_ptr(mask)
        with self._guard():
            rc = _abi.lib.lmaze_v1_set_foveal_goal(self._pp, self._p_layouts, t.data_ptr(), m_ptr, self._pb,
                                                   self.num_envs, self._stream())
        _abi.check("lmaze_v1_set_foveal_goal", rc)
        return self.obs

    def planner_step(self, goal, mask=None):
        """v5/v6 plannerStep(goal) (lmaze_env_v5.py:158-182); goal int[N] in 0..24.  Returns obs_local."""
        t = self._as_i32(goal, self.num_envs)
        m, m_ptr = self._mask_ptr(mask)
        with self._guard():
            rc = _abi.lib.lmaze_v5_planner_step(self._pp, self._p_layouts, t.data_ptr(), m_ptr, self._pb,
                                                self.num_envs, self._stream())
        _abi.check("lmaze_v5_planner_step", rc)
        return self.obs_local

    def safe_foveal_goal(self):
        """v6 safeFovealGoal() (lmaze_env_v6.py:505-523): int32[N] window cell that is not a wall."""
        out = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        with self._guard():
            rc = _abi.lib.lmaze_v6_safe_foveal_goal(self._pp, self._p_layouts, self.seed & (2 ** 64 - 1), self._epoch,
                                                    self.env_base, self._pb, out.data_ptr(), self.num_envs,
                                                    self._stream())
        _abi.check("lmaze_v6_safe_foveal_goal", rc)
        self._epoch += 1
        return out

    def expanded_local(self):
        """(4,35,35) reference layout of obs_local (lmaze_env_v5.py:373-379)."""
        N, E = self.num_envs, self.expansion
        if self._expanded_local is None:
            self._expanded_local = torch.empty((N, 4, 5 * E, 5 * E), dtype=torch.float32, device=self.device)
        with self._guard():
            rc = _abi.lib.lmaze_expand_planes(self.obs_local.data_ptr(), 4, _abi.FOVEA, E,
                                              self._expanded_local.data_ptr(), N, self._stream())
        _abi.check("lmaze_expand_planes", rc)
        return self._expanded_local

    def set_state(self, **kw):
        for name, src in kw.items():
            dst = {"done": self._done_u8, "foveal_done": self._fdone_u8}.get(name, getattr(self, name))
            t = src if isinstance(src, torch.Tensor) else torch.as_tensor(np.asarray(src))
            dst.copy_(t.to(device=self.device).to(dst.dtype).reshape(dst.shape))

    def expanded(self, out=None):
        """(C, 5E, 5E) reference layout of the current obs (the x7 loop, lmaze_env_v2.py:197-203)."""
        N, Cn, E = self.num_envs, self.channels, self.expansion
        if out is None:
            if self._expanded is None:
                self._expanded = torch.empty((N, Cn, 5 * E, 5 * E), dtype=torch.float32, device=self.device)
            out = self._expanded
        with self._guard():
            rc = _abi.lib.lmaze_expand_planes(self.obs.data_ptr(), Cn, _abi.FOVEA, E, out.data_ptr(), N, self._stream())
        _abi.check("lmaze_expand_planes", rc)
        return out

    def episode_stats(self, all_ranks=False):
        """Counters over the batch, off the step path (lmaze_episode_stats): {"done", "goal_rewards",
        "done_steps"} -- envs with done set, envs whose reward is the goal reward, step counts summed over
        the done envs.  all_ranks=True sums them over the process group (one small all_reduce).  Synchronises."""
        out = torch.empty(4, dtype=torch.int64, device=self.device)
        with self._guard():
            rc = _abi.lib.lmaze_episode_stats(self._done_u8.data_ptr(), self.reward.data_ptr(), self.step_count.data_ptr(),
                                              None, self.params.reward_goal, self.num_envs, out.data_ptr(), self._stream())
        _abi.check("lmaze_episode_stats", rc)
        if all_ranks:
            from .sharding import sum_over_ranks
            out = sum_over_ranks(out, device=self.device)
        v = out.tolist()
        return {"done": v[0], "goal_rewards": v[1], "done_steps": v[2]}

    def host_state(self, raw=None):
        """Every per-env scalar on the host (numpy views of one copy of the state block; raw = that block
        already fetched by the caller)."""
        h = self._state.cpu().numpy() if raw is None else raw
        base = self._state.data_ptr()
        out = {}
        for name, t, dt in (("ball_xy", self.ball_xy, np.int32), ("goal_xy", self.goal_xy, np.int32),
                            ("fgoal_xy", self.fgoal_xy, np.int32), ("layout_id", self.layout_id, np.int32),
                            ("step_count", self.step_count, np.int32),
                            ("foveal_step_count", self.foveal_step_count, np.int32),
                            ("reward", self.reward, np.float32), ("foveal_reward", self.foveal_reward, np.float32),
                            ("done", self._done_u8, np.uint8), ("foveal_done", self._fdone_u8, np.uint8),
                            ("ball1_xy", self.ball1_xy, np.int32), ("fovea_xy", self.fovea_xy, np.int32),
                            ("last_xy", self.last_xy, np.int32), ("foveal_goal", self.foveal_goal, np.int32)):
            off = t.data_ptr() - base
            out[name] = h[off:off + t.numel() * t.element_size()].view(dt).reshape(tuple(t.shape))
        return out
