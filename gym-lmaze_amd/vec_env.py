"""LmazeVecEnv: N independent L-mazes held as struct-of-arrays torch tensors in HBM and
stepped by one HIP kernel per step() through the C ABI of include/lmaze.h.

Host side only: buffer ownership, argument marshalling, stream selection.  All arithmetic
of the path (collision check, position update, reward/done, plane render, xE render,
masked reset) runs in liblmaze_hip.so; there is no CPU implementation in this package.

Reference semantics: gym_lmaze/envs/lmaze_env.py (v0) and lmaze_env_v3.py (v3).
"""
import ctypes as C
import logging

import numpy as np
import torch

from . import _abi
from . import layouts as L

_log = logging.getLogger("gym_lmaze_amd")

# per-variant constants the reference hard-codes in __init__
VARIANTS = {
    # lmaze_env.py:16-25, planes lmaze_env.py:208-215 (ball, wall, goal, blank)
    "v0": dict(id=_abi.VARIANT_V0, layout=L.V0_GRID_12, expansion=7, step_limit=100,
               rewards=(-1.0, -0.01, 100.0), channel_mask=(_abi.OBS_BALL, _abi.OBS_WALL, _abi.OBS_GOAL, _abi.OBS_FREE),
               n_actions=4),
    # lmaze_env_v3.py:76-99, planes lmaze_env_v3.py:291-293 (free, ball, goal)
    "v3": dict(id=_abi.VARIANT_V3, layout=L.V3_GRID_18, expansion=4, step_limit=100,
               rewards=(-1.0, -0.01, 100.0), channel_mask=(_abi.OBS_FREE, _abi.OBS_BALL, _abi.OBS_GOAL),
               n_actions=4),
}


def _align(n, a=256):
    return (n + a - 1) // a * a


def resolve_device(device):
    if device is None:
        device = "cuda"
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("gym-lmaze_amd runs on MI355X only (device=%r): the HIP kernels are the "
                           "only implementation of the step path, there is no CPU fallback" % (device,))
    if not torch.cuda.is_available():
        raise RuntimeError("gym-lmaze_amd: no HIP device visible; the step path cannot run")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


class LmazeVecEnv(object):
    """N mazes with `variant` transition rules.

    layout            one layout for every env: row strings / char array / uint8[G,G]
                      (default: the reference's shipped layout for the variant)
    per_env_layouts   uint8[N,G,G] (numpy or torch): every env has its own maze
    env_base          global index of local env 0 when the batch is one shard of a larger
                      one (keys the reset draws; see include/lmaze.h lmaze_reset)
    obs_dtype         "int32" (default; the compact planes BASELINE's metric is quoted on) or "u8": the same LMAZE_OBS_* bit
                      mask in one byte per cell, uint8[N,G,G], 37 + G*G bytes per env-step instead of 37 + 4 G*G (shared
                      layouts, G >= 4; lmaze_step_u8 -- no launch-policy knobs, no one-launch rollout)
    online_autotune   OPT-IN (default False: the library's default launch policy, nothing timed).  True, on
                      large shared-layout batches only (the streaming regime): time the launch policies on
                      the caller's own first ~200 steps, in the caller's own loop -- after 100 untimed steps they cycle
                      through the 8 policies of ONLINE_CANDIDATES, 12 samples each, each launch bracketed by an event
                      pair; a step under a losing policy can take up to twice as long (76 against 152 us for (2, 1) at
                      1M x 11x11) -- and keep the fastest, the library default unless another beats it by more than 1.5 %
                      (see OnlineTuner; `tuning_progress()` reports where it is, `tuned_policy` the winner; one
                      log line when it starts and one when it ends); autotune() or set_launch_policy() switch it off.
                      Results never depend on the policy.
    """

    def __init__(self, num_envs, variant="v0", layout=None, per_env_layouts=None, device=None,
                 expansion=None, step_limit=None, rewards=None, seed=0, env_base=0, validate=True,
                 online_autotune=False, obs_dtype="int32"):
        if variant not in VARIANTS:
            raise ValueError("unknown variant %r (have %s)" % (variant, sorted(VARIANTS)))
        spec = VARIANTS[variant]
        self.variant = variant
        self.num_envs = int(num_envs)
        if self.num_envs < 1:
            raise ValueError("num_envs must be >= 1")
        self.device = resolve_device(device)
        self.expansion = int(expansion if expansion is not None else spec["expansion"])
        self.step_limit = int(step_limit if step_limit is not None else spec["step_limit"])
        self.rewards = tuple(float(r) for r in (rewards if rewards is not None else spec["rewards"]))
        self.channel_mask = spec["channel_mask"]
        self.seed = int(seed)
        self.env_base = int(env_base)
        self._epoch = 0
        self._epoch_words = None        # device-resident epoch pair, allocated by the first captured rollout
        self.tuned_policy = None        # (per_cu, chunks) once autotune() or the online tuner has chosen
        self.placement = None           # autotune(placement_trials=K): where the observation buffer ended up
        self._is_v3 = variant == "v3"

        N = self.num_envs
        if per_env_layouts is not None:
            lay = per_env_layouts
            if not isinstance(lay, torch.Tensor):
                lay = torch.from_numpy(L.to_codes(np.asarray(lay)))
            if lay.dtype != torch.uint8 or lay.dim() != 3 or lay.shape[0] != N or lay.shape[1] != lay.shape[2]:
                raise ValueError("per_env_layouts must be uint8[N,G,G]")
            lay = lay.to(self.device).contiguous()
            if validate:
                _validate_on_device(lay, need_goal=not self._is_v3)
            self.layout_mode = _abi.LAYOUT_PER_ENV
            self.layout = lay
        else:
            codes = L.to_codes(layout if layout is not None else spec["layout"])
            if codes.ndim != 2:
                raise ValueError("layout must be [G,G]; use per_env_layouts for [N,G,G]")
            if validate:
                L.validate(codes, need_goal_marker=not self._is_v3)
            self.layout_mode = _abi.LAYOUT_SHARED
            self.layout = torch.from_numpy(codes.copy()).to(self.device)
        self.grid = int(self.layout.shape[-1])
        if not (3 <= self.grid <= _abi.MAX_GRID):
            raise ValueError("grid side must be in [3, %d]" % _abi.MAX_GRID)
        G = self.grid

        # one allocation for all per-env scalars, so a host mirror is a single copy
        sizes = [("ball_xy", 8 * N), ("goal_xy", 8 * N), ("step_count", 4 * N), ("reward", 4 * N),
                 ("goal_count", 4 * N), ("done", N)]
        offs, total = {}, 0
        for name, sz in sizes:
            offs[name] = total
            total += _align(sz)
        self._state = torch.zeros(total, dtype=torch.uint8, device=self.device)

        def view(name, nbytes, dtype, shape):
            return self._state[offs[name]:offs[name] + nbytes].view(dtype).view(shape)

        self.ball_xy = view("ball_xy", 8 * N, torch.int32, (N, 2))
        self.goal_xy = view("goal_xy", 8 * N, torch.int32, (N, 2))
        self.step_count = view("step_count", 4 * N, torch.int32, (N,))
        self.reward = view("reward", 4 * N, torch.float32, (N,))
        self.goal_count = view("goal_count", 4 * N, torch.int32, (N,))
        self._done_u8 = view("done", N, torch.uint8, (N,))
        self.done = self._done_u8.view(torch.bool)
        if obs_dtype not in ("int32", "u8"):
            raise ValueError("obs_dtype must be 'int32' or 'u8'")
        self._u8 = obs_dtype == "u8"
        if self._u8 and (self.layout_mode != _abi.LAYOUT_SHARED or G < 4):
            raise ValueError("obs_dtype='u8' needs a shared layout with G >= 4")
        self.obs = torch.zeros((N, G, G), dtype=torch.uint8 if self._u8 else torch.int32, device=self.device)
        self._expanded = None

        self.params = _abi.make_params(spec["id"], G, self.layout_mode, self.step_limit, *self.rewards)
        self._pp = C.byref(self.params)
        self._cmask = (C.c_int32 * len(self.channel_mask))(*self.channel_mask)
        self._bind_pointers()
        streaming = self.layout_mode == _abi.LAYOUT_SHARED and N * G * G * 4 > (192 << 20) and not self._u8
        self._tuner = OnlineTuner(self.ONLINE_CANDIDATES) if (online_autotune and streaming) else None
        if self._tuner is not None:
            _log.info("gym-lmaze_amd: online launch-policy tuning on for the next ~%d steps of this %d-env batch",
                      self._tuner.warm + self._tuner.samples * len(self.ONLINE_CANDIDATES), N)

        if not self._is_v3:
            # v0 looks the goal up once from the layout (lmaze_env.py:100-102): first 'X', row-major
            self._fill_goal_from_layout()
        self.reset()

    # ------------------------------------------------------------------ plumbing
    def _bind_pointers(self):
        self._p_layout = self.layout.data_ptr()
        self._p_ball = self.ball_xy.data_ptr()
        self._p_goal = self.goal_xy.data_ptr()
        self._p_step = self.step_count.data_ptr()
        self._p_reward = self.reward.data_ptr()
        self._p_done = self._done_u8.data_ptr()
        self._p_gc = self.goal_count.data_ptr()
        self._p_obs = self.obs.data_ptr()

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _guard(self):
        return torch.cuda.device(self.device)

    def _fill_goal_from_layout(self):
        G = self.grid
        flat = (self.layout.reshape(-1, G * G) == ord("X")).to(torch.int32)
        has = flat.sum(dim=1) > 0
        first = torch.argmax(flat, dim=1).to(torch.int32)
        first = torch.where(has, first, torch.full_like(first, -1))
        gx = torch.div(first, G, rounding_mode="floor")
        gy = first - gx * G
        g = torch.stack([gx, gy], dim=1).to(torch.int32)
        self.goal_xy.copy_(g.expand(self.num_envs, 2) if g.shape[0] == 1 else g)

    def _as_actions(self, actions):
        if isinstance(actions, torch.Tensor):
            a = actions
            if a.device != self.device or a.dtype != torch.int32:
                a = a.to(device=self.device, dtype=torch.int32)
        else:
            a = torch.as_tensor(np.asarray(actions, dtype=np.int64).astype(np.int32), device=self.device)
        a = a.reshape(-1)
        if a.numel() != self.num_envs:
            raise ValueError("expected %d actions, got %d" % (self.num_envs, a.numel()))
        return a.contiguous()

    # ------------------------------------------------------------------ the hot path
    def step(self, actions, render=True, auto_reset=False):
        """One step() of every env.  Returns (obs, reward, done, actions): obs is the compact
        int32[N,G,G] plane buffer (rewritten in place every step), reward float32[N], done
        bool[N].  render=False skips the observation write (transition only).
        auto_reset=True first resets the envs whose done flag is still set from the previous
        step (the user loop `if done: env.reset()`), fused into the same kernel; the result is
        bit-identical to `reset(mask=done)` followed by `step(actions)`."""
        a = self._as_actions(actions)
        with self._guard():
            self._launch_step(a.data_ptr(), self._p_obs if render else None, auto_reset)
        return self.obs, self.reward, self.done, actions

    def step_raw(self, action_ptr, auto_reset=False, epoch_slot=None):
        """step() on a raw device pointer to int32[N] actions (no tensor handling): for
        rollouts over a pre-generated [T,N] action tensor, e.g. under graph capture.  epoch_slot
        (auto_reset under capture): index t of the launch within the captured sequence -- the reset epoch
        then lives on the device (see begin_replay)."""
        self._launch_step(action_ptr, self._p_obs, auto_reset, epoch_slot)

    def _epoch_word_ptrs(self, slot):
        """Device addresses (in, out) of the two alternating epoch words for launch `slot` of a capture."""
        if self._epoch_words is None:
            self._epoch_words = torch.zeros(2, dtype=torch.int64, device=self.device)
        base = self._epoch_words.data_ptr()
        return base + 8 * (slot & 1), base + 8 * ((slot + 1) & 1)

    def begin_replay(self, n_launches):
        """Call before replaying a captured sequence of n_launches auto-reset steps: hands the host's epoch
        count to the device word the first launch reads (one tiny fill on the current stream, no sync) and
        reserves n_launches epochs, so every replay -- and every eager call in between -- draws placements
        no earlier launch has used."""
        self._epoch_word_ptrs(0)
        self._epoch_words[0:1].fill_(self._epoch)
        self._epoch += int(n_launches)

    def tuning_progress(self):
        """None when no online tuning is running, else (timed launches collected, launches needed)."""
        t = self._tuner
        if t is None:
            return None
        return sum(len(v) for v in t.timings.values()), t.samples * len(t.candidates)

    def set_launch_policy(self, per_cu, chunks=1):
        """Fix the launch policy (workgroups per CU, chunks per workgroup) and stop any tuning.  On-die 8x8 batches (up
        to 192 MiB of planes, shared layout) run the wave-autonomous kernel, which reads the same two fields as waves
        per workgroup (1, 2, 4) and envs per wave (1: 64, 2: 32, 3: 16) -- include/lmaze.h, launch_hint; a value
        outside those sets means its default there.  `_abi.describe_step(env.params, N)` shows what a hint selects."""
        self.params.launch_hint = self.launch_hint_of(per_cu, chunks)
        self._tuner = None

    def _launch_step(self, action_ptr, obs_ptr, auto_reset, epoch_slot=None):
        tuner = self._tuner
        if tuner is not None and obs_ptr is not None and not torch.cuda.is_current_stream_capturing():
            cand = tuner.next_candidate()
            self.params.launch_hint = self.launch_hint_of(*cand)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._launch_step_raw(action_ptr, obs_ptr, auto_reset, epoch_slot)
            e1.record()
            best = tuner.add(cand, e0, e1)
            if best is not None:                     # every candidate has its samples: keep the fastest
                self.params.launch_hint = self.launch_hint_of(*best)
                self.tuned_policy, self._tuner = best, None
                _log.info("gym-lmaze_amd: online tuning done, launch policy (workgroups per CU, chunks) = %s", best)
            return
        self._launch_step_raw(action_ptr, obs_ptr, auto_reset, epoch_slot)

    def _launch_step_raw(self, action_ptr, obs_ptr, auto_reset, epoch_slot=None):
        lib, N, st = _abi.lib, self.num_envs, self._stream()
        if self._u8:
            epoch, e_in, e_out = 0, None, None
            if auto_reset and epoch_slot is None:
                epoch = self._epoch
                self._epoch += 1
            elif auto_reset:
                e_in, e_out = self._epoch_word_ptrs(int(epoch_slot))
            rc = lib.lmaze_step_u8(self._pp, self._p_layout, action_ptr, self._p_ball, self._p_goal if self._is_v3 else None,
                                   self._p_step, self._p_reward, self._p_done, None if self._is_v3 else self._p_gc, obs_ptr, N,
                                   1 if auto_reset else 0, self.seed & (2 ** 64 - 1), epoch, self.env_base, e_in, e_out, st)
            _abi.check("lmaze_step_u8", rc)
            return
        if auto_reset:
            seed = self.seed & (2 ** 64 - 1)
            if epoch_slot is None:
                epoch, e_in, e_out = self._epoch, None, None
                self._epoch += 1
            else:           # the count is on the device: frozen host arguments stay valid for every replay
                epoch = 0
                e_in, e_out = self._epoch_word_ptrs(int(epoch_slot))
            if self._is_v3:
                rc = lib.lmaze_step_v3_autoreset(self._pp, self._p_layout, action_ptr, self._p_ball, self._p_goal,
                                                 self._p_step, self._p_reward, self._p_done, obs_ptr, N,
                                                 seed, epoch, self.env_base, e_in, e_out, st)
            else:
                rc = lib.lmaze_step_v0_autoreset(self._pp, self._p_layout, action_ptr, self._p_ball, self._p_step,
                                                 self._p_reward, self._p_done, self._p_gc, obs_ptr, N,
                                                 seed, epoch, self.env_base, e_in, e_out, st)
        elif self._is_v3:
            rc = lib.lmaze_step_v3(self._pp, self._p_layout, action_ptr, self._p_ball, self._p_goal,
                                   self._p_step, self._p_reward, self._p_done, obs_ptr, N, st)
        else:
            rc = lib.lmaze_step_v0(self._pp, self._p_layout, action_ptr, self._p_ball, self._p_step,
                                   self._p_reward, self._p_done, self._p_gc, obs_ptr, N, st)
        _abi.check("lmaze_step_" + self.variant, rc)

    # launch policies autotune() tries: (workgroups per CU, chunks per workgroup) -> LmazeParams.launch_hint
    DEFAULT_POLICY = (0, 0)       # launch_hint = 0: the library's per-shape default (lmaze_step.hip launch_shared)
    # a third element selects the envs per workgroup where the kernel offers a choice (11x11, 12x12: 1 = 64, 2 = 32, 3 = 16;
    # 14x14, 18x18: 1 = 32, 2 = 16; large 8x8 batches: 1 = 128, 2 = 64; 32x32: 1 = 8, 2 = 4; any other G: 1 = 256, 2 = 64,
    # 3 = 16 -- include/lmaze.h)
    CANDIDATES = ((0, 0), (2, 1), (2, 2), (3, 1), (3, 2), (4, 1), (4, 2), (5, 2), (6, 2), (7, 2), (8, 1), (8, 2),
                  (6, 1, 2), (8, 1, 2), (5, 1, 2), (4, 2, 2), (3, 1, 2), (4, 1, 2), (3, 2, 2), (4, 1, 1), (3, 2, 1), (2, 1, 1), (2, 2, 1), (8, 1, 3), (5, 2, 3))

    # the online tuner's own, shorter list (it runs inside the caller's loop): the default and the pairs that have won on
    # some box for some shape (profiles/r02/shape_sweep.jsonl)
    ONLINE_CANDIDATES = ((0, 0), (3, 1), (3, 2), (5, 2), (8, 1), (6, 1, 2), (8, 2, 2), (5, 2, 2))

    @staticmethod
    def launch_hint_of(per_cu, chunks=1, epb_sel=0, no_stagger=False):
        """LmazeParams.launch_hint for `per_cu` workgroups per CU, `chunks` chunks per workgroup and, where the kernel
        offers the choice, the envs-per-workgroup selector (include/lmaze.h: bits 10-11); no_stagger: bit 9."""
        return (int(per_cu) & 15) | ((int(chunks) & 15) << 4) | ((int(epb_sel) & 3) << 10) | (0x200 if no_stagger else 0)

    def autotune(self, auto_reset=False, actions=None, steps=24, candidates=None, warm=150, between=None, rounds=3,
                 placement_trials=0):
        """Pick the launch policy (LmazeParams.launch_hint: workgroups per CU, chunks per workgroup) by
        timing real steps with HIP events; the state is snapshotted and restored, so results are unaffected.
        The optimum is narrow and depends on shape, device and -- most of all -- on WHERE THE INPUTS COME
        FROM: the policy that wins when actions and state sit in the Infinity Cache (3 workgroups per CU)
        loses a third of its rate when the actions are a fresh row from HBM every step (lmaze_step.hip
        launch_shared).  So pass the action tensor the rollout will use (`actions`: int32[T,N] on the
        device; the rows are cycled exactly as rollout() would); without one, a private ring of rows larger
        than the cache is generated, the conservative assumption.  `between`: a callable that enqueues, on the
        current stream, whatever runs between two steps in the real loop (the policy's forward pass): back to
        back, consecutive step launches overlap head to tail and find their state in the cache, and 3
        workgroups per CU win; with half a gigabyte of other traffic in between, that policy took 115 us per
        step instead of 86 and 8 per CU took 93 (tools/evict_study.py) -- then each step is timed on its
        own with an event pair and the median counts.  `warm` untimed launches come first: a cold device
        (the first ~100 launches of a process) ranks the candidates differently from the steady state.
        `rounds` interleaved passes over the candidates, the MEDIAN of a candidate's passes counts, and the library
        default (candidate (0, 0) = launch_hint 0, a per-shape pair) is kept unless another pair beats it by more than
        1.5 %: with the minimum of two short passes
        (round 1) a pair that is fast in a burst and slower sustained could win -- (4, 1) measured 82.6 us while
        tuning and 88.6 us over the 300 timed steps that followed, next to 83.4 for the default.
        placement_trials=K (K > 1): before the policies are timed, K - 1 further observation buffers are allocated and
        the step is timed on each with the (5, 2) policy; the fastest becomes the storage of `self.obs` (the SAME tensor
        object: references the caller holds stay valid), the others are freed.  Where the driver placed the 500-MB write
        target is worth 3-5 % at the default policy and up to 20 % under a capped one.  Round 3 measured what differs
        (tools/placement_pmc.py under rocprofv3 --pmc, LAB_NOTES.md R3.2): NOT address translation (UTCL1 misses 0.06 % of
        requests on fast and slow buffers alike) but the memory side -- 25 % more DRAM write-credit stall cycles
        (TCC_EA0_WRREQ_DRAM_CREDIT_STALL) on the slow allocations, i.e. which channels / banks the buffer's physical pages
        load; a 2-MiB-aligned arena shows the same spread.  `self.placement` records the trial times and, under the policy
        finally chosen, the first allocation's time beside the kept one's (bench.py: roofline.frac_first_allocation).
        Returns {(per_cu, chunks): ms per step}.  Only the shared-layout kernel has these knobs."""
        obs_bytes = self.num_envs * self.grid * self.grid * 4
        if self.layout_mode != _abi.LAYOUT_SHARED or obs_bytes <= (192 << 20) or self._u8:
            return {}       # the knobs only pay in the streaming (non-temporal store) regime; the u8 kernel has none
        cands = [tuple(c) if isinstance(c, (tuple, list)) else (int(c), 1) for c in (candidates or self.CANDIDATES)]
        N = self.num_envs
        if actions is None:
            rows = max(2, min(512, (320 << 20) // (4 * N) + 1))          # > 256 MiB of action rows
            actions = torch.randint(0, 4, (rows, N), dtype=torch.int32, device=self.device)
        elif not (isinstance(actions, torch.Tensor) and actions.dtype == torch.int32 and actions.dim() == 2
                  and actions.shape[1] == N and actions.device == self.device and actions.is_contiguous()):
            raise ValueError("autotune(actions=...) wants a contiguous int32[T,N] tensor on %s" % (self.device,))
        base, stride, R = actions.data_ptr(), N * 4, int(actions.shape[0])
        if int(placement_trials) > 1 and getattr(self, "_captured", 0):
            raise RuntimeError("autotune(placement_trials > 1) would move the observation buffer under %d captured rollout(s), "
                               "which keep raw pointers to it: tune before capture_rollout()" % self._captured)
        snap, epoch = self._state.clone(), self._epoch
        self._tuner = None                   # an explicit autotune replaces the online one
        timings, t = {}, 0
        with self._guard():
            for _ in range(int(warm)):
                self._launch_step(base + (t % R) * stride, self._p_obs, auto_reset)
                t += 1
            if int(placement_trials) > 1:
                bufs = [self.obs] + [torch.empty_like(self.obs) for _ in range(int(placement_trials) - 1)]
                self.params.launch_hint = self.launch_hint_of(5, 2)      # the policy that tells the placements apart
                ms_of = []
                for b in bufs:
                    ptr = b.data_ptr()
                    for _ in range(3):
                        self._launch_step(base + (t % R) * stride, ptr, auto_reset)
                        t += 1
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(12):
                        self._launch_step(base + (t % R) * stride, ptr, auto_reset)
                        t += 1
                    e1.record()
                    e1.synchronize()
                    ms_of.append(e0.elapsed_time(e1) / 12)
                keep = min(range(len(bufs)), key=lambda i: ms_of[i])
                self.placement = {"trials_ms": [round(m, 5) for m in ms_of], "kept": keep}
                first_alloc = self.obs.view_as(self.obs) if keep != 0 else None    # keeps the first allocation alive (timed below)
                if keep != 0:
                    # the SAME tensor object takes over the winning allocation: references the caller already holds to
                    # env.obs stay valid; the first allocation is freed with the rest
                    self.obs.set_(bufs[keep])
                self._p_obs = self.obs.data_ptr()
                self._expanded = None
                del bufs
            for _round in range(int(rounds)):          # interleaved passes; the median of a candidate's passes counts
                for c in cands:
                    self.params.launch_hint = self.launch_hint_of(*c)
                    self._launch_step(base + (t % R) * stride, self._p_obs, auto_reset)   # first launch of a new shape
                    t += 1
                    if between is None:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(steps):
                            self._launch_step(base + (t % R) * stride, self._p_obs, auto_reset)
                            t += 1
                        e1.record()
                        e1.synchronize()
                        ms = e0.elapsed_time(e1) / steps
                    else:
                        pairs = []
                        for _ in range(steps):
                            between()
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            self._launch_step(base + (t % R) * stride, self._p_obs, auto_reset)
                            e1.record()
                            t += 1
                            pairs.append((e0, e1))
                        pairs[-1][1].synchronize()
                        d = sorted(x.elapsed_time(y) for x, y in pairs)
                        ms = d[len(d) // 2]
                    timings.setdefault(c, []).append(ms)
            timings = {c: sorted(v)[len(v) // 2] for c, v in timings.items()}
            best = min(timings, key=timings.get)
            if self.DEFAULT_POLICY in timings and timings[best] > 0.985 * timings[self.DEFAULT_POLICY]:
                best = self.DEFAULT_POLICY
            if int(placement_trials) > 1:
                # what a caller who never tries placements gets: the FIRST allocation under the policy just chosen, timed
                # beside the kept one (bench.py prints both roofline fractions)
                self.params.launch_hint = self.launch_hint_of(*best)
                pair = {}
                for name, ptr in (("kept_ms_tuned", self._p_obs),
                                  ("first_ms_tuned", first_alloc.data_ptr() if first_alloc is not None else self._p_obs)):
                    for _ in range(3):
                        self._launch_step(base + (t % R) * stride, ptr, auto_reset)
                        t += 1
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(steps):
                        self._launch_step(base + (t % R) * stride, ptr, auto_reset)
                        t += 1
                    e1.record()
                    e1.synchronize()
                    pair[name] = round(e0.elapsed_time(e1) / steps, 5)
                # ... and under the LIBRARY DEFAULT policy (launch_hint 0), which is what LmazeVecEnv(...) without any tuning runs
                self.params.launch_hint = 0
                ptr = first_alloc.data_ptr() if first_alloc is not None else self._p_obs
                for _ in range(3):
                    self._launch_step(base + (t % R) * stride, ptr, auto_reset)
                    t += 1
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(steps):
                    self._launch_step(base + (t % R) * stride, ptr, auto_reset)
                    t += 1
                e1.record()
                e1.synchronize()
                pair["first_ms_default"] = round(e0.elapsed_time(e1) / steps, 5)
                self.placement.update(pair)
                first_alloc = None
            self._state.copy_(snap)
            self._epoch = epoch
        best = min(timings, key=timings.get)
        if self.DEFAULT_POLICY in timings and timings[best] > 0.985 * timings[self.DEFAULT_POLICY]:
            best = self.DEFAULT_POLICY
        self.params.launch_hint = self.launch_hint_of(*best)
        self.tuned_policy = best
        self.observe()
        return timings

    def rollout(self, actions, auto_reset=True, device_epoch=False, trajectory=False):
        """T steps over a device tensor int32[T,N] of actions, one kernel per step, no host
        sync (capture_rollout() records it into a hipGraph for launch-bound batch sizes).
        device_epoch: keep the reset epoch on the device (what capture_rollout uses; bit-identical to the
        host-counted epochs when begin_replay(T) precedes it).  Returns the final (obs, reward, done); trajectory=True adds
        every step's reward float32[T,N] and done bool[T,N].  The whole rollout is ONE launch (shared and per-env
        layouts, any batch size; the envs' state stays in registers across the T steps) (include/lmaze.h
        lmaze_rollout)."""
        if not (isinstance(actions, torch.Tensor) and actions.dtype == torch.int32 and actions.dim() == 2
                and actions.shape[1] == self.num_envs and actions.device == self.device and actions.is_contiguous()):
            raise ValueError("rollout() wants a contiguous int32[T,N] tensor on %s" % (self.device,))
        base, stride = actions.data_ptr(), self.num_envs * 4
        T, N = int(actions.shape[0]), self.num_envs
        if not device_epoch and self._tuner is None and not self._u8:
            # lmaze_rollout: ONE launch (the envs' state stays in registers across the T steps); bit-identical to T
            # step() calls
            rew_t = torch.empty((T, N), dtype=torch.float32, device=self.device) if trajectory else None
            done_t = torch.empty((T, N), dtype=torch.uint8, device=self.device) if trajectory else None
            with self._guard():
                rc = _abi.lib.lmaze_rollout(self._pp, self._p_layout, base, T, self._p_ball,
                                            self._p_goal if self._is_v3 else None, self._p_step, self._p_reward, self._p_done,
                                            None if self._is_v3 else self._p_gc, self._p_obs,
                                            rew_t.data_ptr() if trajectory else None, done_t.data_ptr() if trajectory else None,
                                            N, 1 if auto_reset else 0, self.seed & (2 ** 64 - 1), self._epoch, self.env_base,
                                            self._stream())
            _abi.check("lmaze_rollout", rc)
            if auto_reset:
                self._epoch += T
            if trajectory:
                return self.obs, self.reward, self.done, rew_t, done_t.view(torch.bool)
            return self.obs, self.reward, self.done
        if trajectory:
            raise ValueError("rollout(trajectory=True) is not available with a device-resident epoch or while the online tuner runs")
        with self._guard():
            for t in range(actions.shape[0]):
                self._launch_step(base + t * stride, self._p_obs, auto_reset, t if device_epoch else None)
        return self.obs, self.reward, self.done

    def observe(self, mask_ptr=None):
        """Re-render the compact planes of the current state (no transition)."""
        if self._u8:
            with self._guard():
                rc = _abi.lib.lmaze_observe_u8(self._pp, self._p_layout, self._p_ball, self._p_goal if self._is_v3 else None,
                                               mask_ptr, self._p_obs, self.num_envs, self._stream())
            _abi.check("lmaze_observe_u8", rc)
            return self.obs
        with self._guard():
            rc = _abi.lib.lmaze_observe(self._pp, self._p_layout, self._p_ball,
                                        self._p_goal if self._is_v3 else None, self._p_obs,
                                        self.num_envs, self._stream())
        _abi.check("lmaze_observe", rc)
        return self.obs

    def reset(self, mask=None, seed=None):
        """Masked on-device reset (mask: bool/uint8[N], None = all).  Returns the compact obs."""
        if seed is not None:
            self.seed = int(seed)
            self._epoch = 0
        m_ptr = None
        if mask is not None:
            m = mask if isinstance(mask, torch.Tensor) else torch.as_tensor(np.asarray(mask), device=self.device)
            m = m.to(device=self.device)
            m = (m.view(torch.uint8) if m.dtype == torch.bool else (m != 0).to(torch.uint8)).contiguous()
            if m.numel() != self.num_envs:
                raise ValueError("mask must have %d entries" % self.num_envs)
            m_ptr = m.data_ptr()
        with self._guard():
            rc = _abi.lib.lmaze_reset(self._pp, self._p_layout, m_ptr, self.seed & (2 ** 64 - 1), self._epoch,
                                      self.env_base, self._p_ball, self._p_goal if self._is_v3 else None,
                                      self._p_step, self._p_reward, self._p_done, None if self._u8 else self._p_obs,
                                      self.num_envs, self._stream())
        _abi.check("lmaze_reset", rc)
        self._epoch += 1
        if self._u8:
            self.observe(mask_ptr=m_ptr)       # the narrow planes of the envs that were reset
        return self.obs

    def set_state(self, ball_xy=None, goal_xy=None, step_count=None, reward=None, goal_count=None, done=None):
        """Inject state (placement chosen by the caller, e.g. the reference's own RNG stream)."""
        for dst, src in ((self.ball_xy, ball_xy), (self.goal_xy, goal_xy), (self.step_count, step_count),
                         (self.reward, reward), (self.goal_count, goal_count), (self._done_u8, done)):
            if src is not None:
                t = src if isinstance(src, torch.Tensor) else torch.as_tensor(np.asarray(src))
                dst.copy_(t.to(device=self.device).to(dst.dtype).reshape(dst.shape))

    def expanded(self, out=None):
        """Reference-layout observation float32[N,C,G*E,G*E] of the current compact planes
        (the upsample loop of lmaze_env.py:217-234)."""
        N, G, E, Cn = self.num_envs, self.grid, self.expansion, len(self.channel_mask)
        if out is None:
            if self._expanded is None:
                self._expanded = torch.empty((N, Cn, G * E, G * E), dtype=torch.float32, device=self.device)
            out = self._expanded
        src = self.obs.to(torch.int32) if self._u8 else self.obs      # the x E render reads int32 planes
        with self._guard():
            rc = _abi.lib.lmaze_render_expanded(src.data_ptr(), G, E, self._cmask, Cn, out.data_ptr(), N,
                                                self._stream())
        _abi.check("lmaze_render_expanded", rc)
        return out

    def capture_rollout(self, actions, auto_reset=False):
        """Capture the T = actions.shape[0] launches of rollout(actions) into ONE hipGraph and return it
        (a RolloutGraph; call .replay()).  For launch-bound batch sizes (65 536 x 8x8 is 6 us per
        step, a third of it launch gap).  The launches allocate nothing and never synchronise, so they are
        capturable as they are.  With auto_reset the reset epoch is a device word the launches hand on to
        each other (lmaze_step_*_autoreset, epoch_in_dev / epoch_out_dev), so every replay draws fresh
        placements, and exactly those the same steps launched eagerly would draw."""
        T = int(actions.shape[0])
        if self._tuner is not None:             # still cycling through candidates: a graph bakes the default policy
            self.params.launch_hint = 0
        if auto_reset:
            self._epoch_word_ptrs(0)            # allocate before capture
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                self.rollout(actions, auto_reset=auto_reset, device_epoch=auto_reset)
        torch.cuda.current_stream(self.device).wait_stream(side)
        self._captured = getattr(self, "_captured", 0) + 1      # the graph keeps raw pointers: autotune() no longer moves obs
        return RolloutGraph(self, graph, T, auto_reset)

    def episode_stats(self, all_ranks=False):
        """Counters over the batch, off the step path: {"done", "goal_rewards", "done_steps", "goal_count"}.
        all_ranks=True sums them over the process group (one all_reduce of four int64 over RCCL): the only
        collective the library ever issues.  Synchronises (returns Python ints)."""
        out = torch.empty(4, dtype=torch.int64, device=self.device)
        with self._guard():
            rc = _abi.lib.lmaze_episode_stats(self._p_done, self._p_reward, self._p_step,
                                              None if self._is_v3 else self._p_gc, self.rewards[2], self.num_envs,
                                              out.data_ptr(), self._stream())
        _abi.check("lmaze_episode_stats", rc)
        if all_ranks:
            from .sharding import sum_over_ranks
            out = sum_over_ranks(out, device=self.device)
        v = out.tolist()
        return {"done": v[0], "goal_rewards": v[1], "done_steps": v[2], "goal_count": v[3]}

    def planes(self, out=None):
        """The reference's unexpanded planes float32[N,C,G,G] (what it calls retState, lmaze_env.py:208-215):
        the x1 case of the expanded render, ready as network input."""
        N, G, Cn = self.num_envs, self.grid, len(self.channel_mask)
        if out is None:
            out = torch.empty((N, Cn, G, G), dtype=torch.float32, device=self.device)
        with self._guard():
            rc = _abi.lib.lmaze_render_expanded(self._p_obs, G, 1, self._cmask, Cn, out.data_ptr(), N, self._stream())
        _abi.check("lmaze_render_expanded", rc)
        return out

    def host_state(self, raw=None):
        """One device->host copy of every per-env scalar; returns numpy views.  raw: bytes of the state block
        already on the host (uint8 array the size of `_state`), parsed instead of copying again."""
        h = self._state.cpu().numpy() if raw is None else raw
        base = self._state.data_ptr()

        def v(t, dtype, shape):
            off = t.data_ptr() - base
            return h[off:off + t.numel() * t.element_size()].view(dtype).reshape(shape)

        N = self.num_envs
        return dict(ball_xy=v(self.ball_xy, np.int32, (N, 2)), goal_xy=v(self.goal_xy, np.int32, (N, 2)),
                    step_count=v(self.step_count, np.int32, (N,)), reward=v(self.reward, np.float32, (N,)),
                    goal_count=v(self.goal_count, np.int32, (N,)), done=v(self._done_u8, np.uint8, (N,)))


class OnlineTuner:
    """Launch-policy selection on the caller's own steps (LmazeVecEnv, streaming regime).  The best (workgroups
    per CU, chunks per workgroup) pair depends on the device and on what else runs between two steps -- with
    the per-env state still cached, 3 per CU wins; after half a gigabyte of other traffic, 8 per CU does
     -- so instead of guessing, the first launches of a run cycle through the candidates, each
    timed on its own with an event pair (no synchronisation: finished pairs are collected as they complete),
    and once every candidate has `samples` timings the lowest median is kept.  `warm` launches are ignored first
    (a cold device ranks differently).  Results never depend on the policy, only the pace of those launches."""

    def __init__(self, candidates, warm=100, samples=12):
        self.candidates = [tuple(c) for c in candidates]
        self.warm, self.samples = int(warm), int(samples)
        self.timings = {c: [] for c in self.candidates}
        self._pending, self._count = [], 0

    def next_candidate(self):
        return self.candidates[self._count % len(self.candidates)]

    def add(self, cand, e0, e1):
        """Register one timed launch; returns the winner once every candidate has enough samples."""
        self._count += 1
        if self._count > self.warm:
            self._pending.append((cand, e0, e1))
        while self._pending and self._pending[0][2].query():
            c, a, b = self._pending.pop(0)
            self.timings[c].append(a.elapsed_time(b))
        if all(len(v) >= self.samples for v in self.timings.values()):
            med = {c: sorted(v)[len(v) // 2] for c, v in self.timings.items()}
            best = min(med, key=med.get)
            # as autotune(): the library default stays unless another policy beats it by more than 1.5 %
            if (0, 0) in med and med[best] > 0.985 * med[(0, 0)]:
                best = (0, 0)
            return best
        return None


class RolloutGraph:
    """A captured rollout (LmazeVecEnv.capture_rollout): replay() relaunches its T steps in one go."""

    def __init__(self, env, graph, n_launches, auto_reset):
        self.env, self.graph, self.n_launches, self.auto_reset = env, graph, n_launches, auto_reset

    def replay(self):
        if self.auto_reset:
            self.env.begin_replay(self.n_launches)
        self.graph.replay()


def _validate_on_device(lay, need_goal=True):
    W = ord("W")
    ok = ((lay[:, 0, :] == W).all() & (lay[:, -1, :] == W).all() & (lay[:, :, 0] == W).all()
          & (lay[:, :, -1] == W).all())
    if not bool(ok):
        raise ValueError("every layout needs a full 'W' border")
    codes = torch.tensor([ord(c) for c in "WBSX"], dtype=torch.uint8, device=lay.device)
    if not bool(torch.isin(lay, codes).all()):
        raise ValueError("layout cells must be one of 'W', 'B', 'S', 'X'")
    if need_goal and not bool((lay == ord("X")).flatten(1).any(dim=1).all()):
        raise ValueError("a layout has no 'X' cell")
