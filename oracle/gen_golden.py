#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (oracle/): writes the golden fixtures under tests/golden/.

Runs in the build container only (needs /root/reference).  It executes the reference's
own `reset()`/`step()` (loaded by oracle/ref_loader.py, unmodified) on seeded action
sequences and records, per step, everything the parity tests compare against:

  layout        uint8[G,G]    the reference env's `grid` as ASCII codes ('W','B','S','X')
  E             int           expansionRatio
  actions       int32[T]      action fed to step() (v3: the decimal string of it)
  reset_before  uint8[T]      1 if reset() ran right before step t (t=0: the initial state)
  ball_before   int32[T,2]    (ball_x0, ball_y0) going into step t
  reward        float64[T]    reward exactly as returned (Python float)
  done          uint8[T]
  ball          int32[T,2]    ball after step t
  step_count    int32[T]      env.stepCount after step t
  planes        uint8[T,G,G]  obs[:, ::E, ::E] packed, bit c = reference channel c
  obs_hash      uint64[T]     first 8 bytes (little endian) of sha256(obs.tobytes())
  reset_planes / reset_hash   same two for the obs returned by each reset() (in order)

Before packing, every observation is checked to be 0/1-valued and to equal the exact
ExE nearest-neighbour replication of its compact planes, so (planes, E) determine the
full float32 observation bit-for-bit; obs_hash pins it a second time.

Data only: no reference source text is written anywhere.
"""
import hashlib
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

OUT = os.environ.get("LMAZE_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")


def grid_from_rows(rows):
    return np.array([list(r) for r in rows])


def to_codes(grid):
    return np.vectorize(ord)(grid).astype(np.uint8)


def obs_hash(obs):
    assert obs.dtype == np.float32 and obs.flags["C_CONTIGUOUS"]
    return np.frombuffer(hashlib.sha256(obs.tobytes()).digest()[:8], dtype="<u8")[0]


def pack_planes(obs, E):
    """obs (C, G*E, G*E) float32 0/1 -> uint8 (G, G), bit c = channel c; verifies replication."""
    obs = np.asarray(obs)
    comp = obs[:, ::E, ::E]
    assert np.isin(comp, (0.0, 1.0)).all(), "non 0/1 plane value"
    rep = np.repeat(np.repeat(comp, E, axis=1), E, axis=2)
    assert rep.shape == obs.shape and (rep == obs).all(), "obs is not an exact ExE replication"
    out = np.zeros(comp.shape[1:], dtype=np.uint8)
    for c in range(comp.shape[0]):
        out |= (comp[c] != 0).astype(np.uint8) << c
    return out


# ----------------------------------------------------------------------------------------
# layouts used for the BASELINE sizes
# ----------------------------------------------------------------------------------------
def layout_8x8_bordered():
    """The commented 8x8 literal of lmaze_env.py:28-35 with a 'W' border enforced on the
    last row/column (the literal leaves them open, which would index out of the array)."""
    rows = ["WWWWWWWW",
            "WSBBBWWW",
            "WBWWWWWW",
            "WBWBWBBW",
            "WBWBWBWW",
            "WBBBWXWW",
            "WBWBWBWW",
            "WWWWWWWW"]
    return grid_from_rows(rows)


def layout_open_room(G, goal=None):
    g = np.full((G, G), "B")
    g[0, :] = g[-1, :] = g[:, 0] = g[:, -1] = "W"
    g[1, 1] = "S"
    gx, gy = goal if goal else (G // 2, G // 2)
    g[gx, gy] = "X"
    return g


def layout_11x11_v3_literal():
    """The commented 11x11 literal at lmaze_env_v3.py:61-71 (4-cell pad)."""
    rows = ["WWWWWWWWWWW"] * 4 + ["WWWWSBBWWWW", "WWWWBWBWWWW", "WWWWBWXWWWW"] + ["WWWWWWWWWWW"] * 4
    return grid_from_rows(rows)


def layout_random(G, seed, p_wall=0.25):
    rs = np.random.RandomState(seed)
    g = np.where(rs.rand(G, G) < p_wall, "W", "B")
    g[0, :] = g[-1, :] = g[:, 0] = g[:, -1] = "W"
    free = np.argwhere(g == "B")
    sx, sy = free[rs.randint(len(free))]
    g[sx, sy] = "S"
    free = np.argwhere(g == "B")
    gx, gy = free[rs.randint(len(free))]
    g[gx, gy] = "X"
    return g


# ----------------------------------------------------------------------------------------
# v0  (lmaze_env.py)
# ----------------------------------------------------------------------------------------
def rollout_v0(grid, actions, seed, reset_on_done=True, random_ball=True):
    env = ref_loader.make("v0")
    env.RANDOM_BALL = random_ball            # flipped after construction, as a user does (lmaze_env.py:25,70,82-89)
    if grid is not None:  # re-size the reference by attribute override
        env.grid = grid
        env.realgrid = grid.shape[0]
        env.gridsize = env.realgrid * env.expansionRatio
    E, G = env.expansionRatio, env.realgrid
    random.seed(seed)
    T = len(actions)
    rec = dict(layout=to_codes(env.grid), E=np.int32(E), seed=np.int64(seed), random_ball=np.uint8(random_ball),
               actions=np.asarray(actions, dtype=np.int32),
               reset_before=np.zeros(T, np.uint8), ball_before=np.zeros((T, 2), np.int32),
               reward=np.zeros(T, np.float64), done=np.zeros(T, np.uint8),
               ball=np.zeros((T, 2), np.int32), step_count=np.zeros(T, np.int32),
               goal_count=np.zeros(T, np.int32),
               planes=np.zeros((T, G, G), np.uint8), obs_hash=np.zeros(T, np.uint64))
    reset_planes, reset_hash = [], []
    env.goalCount = 0
    need_reset = True
    for t in range(T):
        if need_reset:
            o = env.reset()
            reset_planes.append(pack_planes(o, E))
            reset_hash.append(obs_hash(o))
            rec["reset_before"][t] = 1
            need_reset = False
        rec["ball_before"][t] = (env.ball_x0, env.ball_y0)
        o, r, d, info = env.step(actions[t])
        assert info == int(actions[t]) and type(r) is float and type(d) is bool
        assert o.dtype == np.float32 and o.shape == (4, G * E, G * E)
        rec["reward"][t] = r
        rec["done"][t] = d
        rec["ball"][t] = (env.ball_x0, env.ball_y0)
        rec["step_count"][t] = env.stepCount
        rec["goal_count"][t] = env.goalCount
        rec["planes"][t] = pack_planes(o, E)
        rec["obs_hash"][t] = obs_hash(o)
        if d and reset_on_done:
            need_reset = True
    rec["goal"] = np.array([env.goal_x, env.goal_y], np.int32)
    rec["reset_planes"] = np.stack(reset_planes)
    rec["reset_hash"] = np.array(reset_hash, np.uint64)
    return rec


# ----------------------------------------------------------------------------------------
# v3  (lmaze_env_v3.py): string actions, random goal, look-ahead goal test
# ----------------------------------------------------------------------------------------
def rollout_v3(actions, seed, mode="train", grid=None, reset_on_done=True, random_ball=True, random_goal=True):
    random.seed(seed)                        # the constructor's own reset() draws too (lmaze_env_v3.py:125)
    env = ref_loader.make("v3")
    env.RANDOM_BALL, env.RANDOM_GOAL = random_ball, random_goal      # lmaze_env_v3.py:100-101,147-164
    if grid is not None:
        env.grid = grid
        env.realgrid = grid.shape[0]
        env.fovea = env.realgrid
        env.gridsize = env.fovea * env.expansionRatio
        env.retStateExpanded = np.zeros((env.stateChannel, env.gridsize, env.gridsize), dtype=np.float32)
    E, G = env.expansionRatio, env.realgrid
    random.seed(seed)
    T = len(actions)
    rec = dict(layout=to_codes(env.grid), E=np.int32(E), seed=np.int64(seed),
               random_ball=np.uint8(random_ball), random_goal=np.uint8(random_goal),
               mode_test=np.uint8(mode == "test"),
               actions=np.asarray(actions, dtype=np.int32),
               reset_before=np.zeros(T, np.uint8), ball_before=np.zeros((T, 2), np.int32),
               goal_before=np.zeros((T, 2), np.int32),
               reward=np.zeros(T, np.float64), done=np.zeros(T, np.uint8),
               ball=np.zeros((T, 2), np.int32), step_count=np.zeros(T, np.int32),
               planes=np.zeros((T, G, G), np.uint8), obs_hash=np.zeros(T, np.uint64))
    reset_planes, reset_hash = [], []
    need_reset = True
    for t in range(T):
        if need_reset:
            o = env.reset(mode)
            reset_planes.append(pack_planes(o, E))
            reset_hash.append(obs_hash(o))
            rec["reset_before"][t] = 1
            need_reset = False
        rec["ball_before"][t] = (env.ball_x0, env.ball_y0)
        rec["goal_before"][t] = (env.goal_x, env.goal_y)
        a = int(actions[t])
        # ids 0..3 go in as their decimal strings (the only spelling the reference moves on,
        # lmaze_env_v3.py:236-247); every other id is fed as a raw int, which is a no-op move.
        arg = str(a) if 0 <= a <= 3 else a
        o, r, d, info = env.step(arg)
        assert info is arg or info == arg
        assert o.dtype == np.float32 and o.shape == (3, G * E, G * E)
        rec["reward"][t] = r
        rec["done"][t] = d
        rec["ball"][t] = (env.ball_x0, env.ball_y0)
        rec["step_count"][t] = env.stepCount
        rec["planes"][t] = pack_planes(o, E)
        rec["obs_hash"][t] = obs_hash(o)
        if d and reset_on_done:
            need_reset = True
    rec["reset_planes"] = np.stack(reset_planes)
    rec["reset_hash"] = np.array(reset_hash, np.uint64)
    return rec


def mixed_actions(seed, T, lo=-1, hi=7):
    """Mostly 0..3, with out-of-range ids sprinkled in (Appendix B-2)."""
    rs = np.random.RandomState(seed)
    a = rs.randint(0, 4, T)
    odd = rs.rand(T) < 0.08
    a[odd] = rs.randint(lo, hi + 1, odd.sum())
    return a.astype(np.int32)


def save(name, rec):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print("wrote %-28s T=%d  %.1f KB" % (name, len(rec["actions"]), os.path.getsize(path) / 1024.0))


def gen_v0():
    # C1: shipped 12x12, the exact action stream SURVEY 8(d) names
    a = np.random.RandomState(123).randint(0, 4, 1000).astype(np.int32)
    save("v0_c1_g12_seed0", rollout_v0(None, a, seed=0))
    save("v0_g12_mixed_seed1", rollout_v0(None, mixed_actions(11, 300), seed=1))
    save("v0_g8_seed0", rollout_v0(layout_8x8_bordered(), mixed_actions(12, 400), seed=0))
    save("v0_g11_open_seed0", rollout_v0(layout_open_room(11, (5, 5)), mixed_actions(13, 400), seed=0))
    save("v0_g11_v3lit_seed2", rollout_v0(layout_11x11_v3_literal(), mixed_actions(14, 300), seed=2))
    save("v0_g32_rand_seed0", rollout_v0(layout_random(32, 7), mixed_actions(15, 120), seed=0))
    # no reset on done: stepping past done (step 100 True, 101 False; sticky 100.0 on 'S')
    save("v0_g12_noreset_seed3", rollout_v0(None, mixed_actions(16, 260), seed=3, reset_on_done=False))


def gen_v3():
    save("v3_g18_seed0", rollout_v3(mixed_actions(31, 400), seed=0))
    save("v3_g18_test_seed1", rollout_v3(mixed_actions(32, 250), seed=1, mode="test"))
    save("v3_g18_noreset_seed2", rollout_v3(mixed_actions(33, 230), seed=2, reset_on_done=False))
    save("v3_g11_open_seed3", rollout_v3(mixed_actions(34, 300), seed=3, grid=layout_open_room(11, (5, 5))))


# ----------------------------------------------------------------------------------------
# foveal variants (5x5 window observations): v1, v2, v4, v5/v6
#   planes   float32[T,C,5,5]   obs[:, ::E, ::E] (v4-v6 carry a float visit-map plane), after the
#                               same exact-replication check
# ----------------------------------------------------------------------------------------
def unexpand(obs, E):
    obs = np.asarray(obs)
    comp = np.ascontiguousarray(obs[:, ::E, ::E]).astype(np.float32)
    rep = np.repeat(np.repeat(comp, E, axis=1), E, axis=2)
    assert obs.dtype == np.float32 and rep.shape == obs.shape and (rep == obs).all(), "not an exact ExE replication"
    return comp


_FIVE = None


def five_layouts():
    """The five 18x18 layouts setGrid() chooses from (lmaze_env_v2.py:309-405), read off a live object."""
    global _FIVE
    if _FIVE is None:
        import numpy.random as npr
        # constructing the env runs its reset(): draws from the global `random` / `np.random` streams.  Put both
        # back afterwards, so that a recording never depends on whether this cache was already filled -- i.e. on
        # which generators ran before it (round 1's v5_seed0 did: VERDICT r01 "what's weak" 2)
        st_py, st_np = random.getstate(), npr.get_state()
        env = ref_loader.make("v2")
        orig = npr.randint
        tabs = []
        try:
            for k in range(1, 6):
                npr.randint = lambda a, b, k=k: k
                env.setGrid()
                tabs.append(to_codes(env.grid))
        finally:
            npr.randint = orig
            random.setstate(st_py)
            npr.set_state(st_np)
        _FIVE = np.stack(tabs)
    return _FIVE


def layout_id_of(grid):
    codes = to_codes(grid)
    for k, t in enumerate(five_layouts()):
        if t.shape == codes.shape and (t == codes).all():
            return k
    raise AssertionError("grid is not one of the five shipped layouts")


def foveal_actions(seed, T):
    return np.random.RandomState(seed).randint(0, 25, T).astype(np.int32)


def rollout_v24(variant, actions, seed, reset_on_done=True, random_ball=True, random_goal=True, record_visit=False):
    """v2 (lmaze_env_v2.py) and v4 (lmaze_env_v4.py): 25-way teleport-in-fovea action.  record_visit (v4): the whole
    visit map state[2] after every step, not only the two windows the observation shows."""
    five_layouts()
    import contextlib
    import io
    random.seed(seed)
    np.random.seed(seed)
    env = ref_loader.make(variant)          # the constructor already ran one reset()
    env.RANDOM_BALL, env.RANDOM_GOAL = random_ball, random_goal      # lmaze_env_v2.py:51-52,277-299
    E, C = env.expansionRatio, env.stateChannel
    T = len(actions)
    rec = dict(E=np.int32(E), seed=np.int64(seed), layouts=five_layouts(),
               random_ball=np.uint8(random_ball), random_goal=np.uint8(random_goal),
               actions=np.asarray(actions, dtype=np.int32),
               reset_before=np.zeros(T, np.uint8), ball_before=np.zeros((T, 2), np.int32),
               goal_before=np.zeros((T, 2), np.int32), layout_id=np.zeros(T, np.int32),
               reward=np.zeros(T, np.float64), done=np.zeros(T, np.uint8),
               ball=np.zeros((T, 2), np.int32), step_count=np.zeros(T, np.int32),
               planes=np.zeros((T, C, 5, 5), np.float32), obs_hash=np.zeros(T, np.uint64))
    if record_visit:
        rec["visit"] = np.zeros((T, 18, 18), np.float32)
    reset_planes, reset_hash = [], []
    need_reset = True
    first = None
    for t in range(T):
        if need_reset:
            o = env.reset()
            reset_planes.append(unexpand(o, E))
            reset_hash.append(obs_hash(o))
            rec["reset_before"][t] = 1
            need_reset = False
        rec["ball_before"][t] = (env.ball_x0, env.ball_y0)
        rec["goal_before"][t] = (env.goal_x, env.goal_y)
        rec["layout_id"][t] = layout_id_of(env.grid)
        with contextlib.redirect_stdout(io.StringIO()):   # v4 prints state[2] on done (lmaze_env_v4.py:269)
            o, r, d, info = env.step(actions[t])
        first = o if first is None else first
        assert o is first and info == int(actions[t]) and o.shape == (C, 5 * E, 5 * E)
        rec["reward"][t] = r
        rec["done"][t] = d
        rec["ball"][t] = (env.ball_x0, env.ball_y0)
        rec["step_count"][t] = env.stepCount
        rec["planes"][t] = unexpand(o, E)
        rec["obs_hash"][t] = obs_hash(o)
        if record_visit:
            rec["visit"][t] = env.state[2]
        if d and reset_on_done:
            need_reset = True
    rec["reset_planes"] = np.stack(reset_planes)
    rec["reset_hash"] = np.array(reset_hash, np.uint64)
    return rec


def deep_decay_actions():
    """A scripted v4 walk WITHOUT resets that leaves one corner of the map alone for 112 ... 170 steps and then returns
    to it: the reference halves the whole float32 plane every step (lmaze_env_v4.py:211-214), so the cells there decay
    through the normal range (exact), the subnormal range (rounded to nearest-even on EVERY step, below 2^-126) and to
    zero (below 2^-149) before they are shown again.  Action 5*i+j teleports by (i-2, j-2), clamped to [2, 15]."""
    rs = np.random.RandomState(91)
    acts = []
    for stay in (112, 125, 131, 136, 142, 150, 170):
        acts += [24] * 8                                   # to the (15, 15) corner
        acts += list(rs.choice([12, 13, 11, 17, 7, 18, 6], 6))   # a few steps there: multi-bit mantissas
        acts += [0] * 8                                    # to the (2, 2) corner
        acts += list(rs.choice([12, 13, 11, 17, 7, 12, 12], stay))
    acts += [24] * 8 + [12] * 4
    return np.array(acts, np.int32)


def rollout_v1(actions, fgoals, seed, reset_on_done=True):
    """v1 (lmaze_env_v1.py): 14x14, 5x5 window, foveal goal set by setFovealGoal(i, j) whenever the
    foveal episode finished (the two-level loop the class is written for)."""
    rs = np.random.RandomState(seed)
    env = ref_loader.make("v1")
    E = env.expansionRatio
    T = len(actions)
    rec = dict(layout=to_codes(env.grid), E=np.int32(E), seed=np.int64(seed),
               actions=np.asarray(actions, dtype=np.int32),
               reset_before=np.zeros(T, np.uint8), setgoal_before=np.zeros(T, np.uint8),
               setgoal_ij=np.zeros((T, 2), np.int32),
               ball_before=np.zeros((T, 2), np.int32), fgoal_before=np.zeros((T, 2), np.int32),
               fstep_before=np.zeros(T, np.int32),
               reward=np.zeros(T, np.float64), foveal_reward=np.zeros(T, np.float64),
               done=np.zeros(T, np.uint8), foveal_done=np.zeros(T, np.uint8),
               ball=np.zeros((T, 2), np.int32), step_count=np.zeros(T, np.int32),
               foveal_step_count=np.zeros(T, np.int32),
               planes=np.zeros((T, 4, 5, 5), np.float32), obs_hash=np.zeros(T, np.uint64),
               setgoal_planes=np.zeros((T, 4, 5, 5), np.float32),
               global_planes=np.zeros((T, 4, 5, 5), np.float32), state_hash=np.zeros(T, np.uint64))
    reset_planes, reset_hash = [], []
    need_reset, need_goal = True, True
    k = 0
    for t in range(T):
        if need_reset:
            o = env.reset()
            reset_planes.append(unexpand(o, E))
            reset_hash.append(obs_hash(o))
            rec["reset_before"][t] = 1
            need_reset, need_goal = False, True
        if need_goal:
            i, j = fgoals[k % len(fgoals)]
            k += 1
            o = env.setFovealGoal(int(i), int(j))
            rec["setgoal_before"][t] = 1
            rec["setgoal_ij"][t] = (i, j)
            rec["setgoal_planes"][t] = unexpand(o, E)
            need_goal = False
        rec["ball_before"][t] = (env.ball_x0, env.ball_y0)
        rec["fgoal_before"][t] = (env.f_goal_x, env.f_goal_y)
        rec["fstep_before"][t] = env.fovealStepCount
        a = int(actions[t])
        o, r, fr, fd, d, info = env.step(a)
        assert info == a and o.shape == (4, 5 * E, 5 * E)
        rec["reward"][t] = r
        rec["foveal_reward"][t] = fr
        rec["done"][t] = d
        rec["foveal_done"][t] = fd
        rec["ball"][t] = (env.ball_x0, env.ball_y0)
        rec["step_count"][t] = env.stepCount
        rec["foveal_step_count"][t] = env.fovealStepCount
        rec["planes"][t] = unexpand(o, E)
        rec["obs_hash"][t] = obs_hash(o)
        # the two extras a caller can ask for between steps (lmaze_env_v1.py:204-238, 289-290); neither changes the env
        gv = env.getGlobalView()
        rec["global_planes"][t] = unexpand(gv, E)
        st4, r0, d0, info0 = env.initState()
        assert st4 is env.state and r0 == env.originalReward and d0 == d and info0 == {'newState': True}
        rec["state_hash"][t] = obs_hash(np.ascontiguousarray(st4, dtype=np.float32))
        if d and reset_on_done:
            need_reset = True
        elif fd:
            need_goal = True
    rec["goal"] = np.array([env.goal_x, env.goal_y], np.int32)
    rec["reset_planes"] = np.stack(reset_planes)
    rec["reset_hash"] = np.array(reset_hash, np.uint64)
    return rec


def gen_v2():
    save("v2_seed0", rollout_v24("v2", foveal_actions(41, 400), seed=0))
    save("v2_seed1", rollout_v24("v2", foveal_actions(42, 300), seed=1))
    save("v2_noreset_seed2", rollout_v24("v2", foveal_actions(43, 150), seed=2, reset_on_done=False))


def gen_v4():
    save("v4_seed0", rollout_v24("v4", foveal_actions(51, 400), seed=0))
    save("v4_seed1", rollout_v24("v4", foveal_actions(52, 300), seed=1))
    save("v4_noreset_seed2", rollout_v24("v4", foveal_actions(53, 150), seed=2, reset_on_done=False))
    save("v4_deepdecay_seed9", rollout_v24("v4", deep_decay_actions(), seed=9, reset_on_done=False, record_visit=True))


def gen_v1():
    rs = np.random.RandomState(61)
    fg = rs.randint(0, 5, (64, 2))
    save("v1_seed0", rollout_v1(mixed_actions(62, 600, lo=-1, hi=5), fg, seed=0))
    # foveal goals kept inside the reachable cross so local goals get hit
    fg2 = np.array([(2, 3), (3, 2), (2, 1), (1, 2), (2, 2), (2, 4), (4, 2)])
    save("v1_seed1", rollout_v1(np.random.RandomState(63).randint(0, 4, 500).astype(np.int32), fg2, seed=1))
    save("v1_noreset_seed2", rollout_v1(mixed_actions(64, 450), fg, seed=2, reset_on_done=False))
    # scripted: walk S(2,2) -> X(6,6) twice (global reward 1.0 / done), some wall bumps in between, with
    # foveal goals chosen one step ahead so the local goal is hit too
    path = [1, 1, 1, 1, 3, 3, 1, 1, 1, 3, 3, 0, 0, 0]
    acts = np.array(path + [0, 2, 7] + path + [0], np.int32)
    fg3 = np.array([(3, 2), (2, 3), (1, 2), (3, 2), (2, 3)])
    save("v1_scripted_goal", rollout_v1(acts, fg3, seed=3))


# ----------------------------------------------------------------------------------------
# v5 / v6 (lmaze_env_v5.py, lmaze_env_v6.py): two-level planner / local loop.  The rollout is a
# sequence of EVENTS (reset, plannerStep(g), step(a)); after every event the full state is recorded.
# ----------------------------------------------------------------------------------------
EV_RESET, EV_PLANNER, EV_STEP = 0, 1, 2


def rollout_v56(variant, n_events, seed, safe_goals=False, calm=False):
    import contextlib
    import io
    rs = np.random.RandomState(seed + 1000)
    five_layouts()                  # before seeding (and it restores the global streams anyway)
    random.seed(seed)
    np.random.seed(seed)
    env = ref_loader.make(variant)
    E = env.expansionRatio
    T = n_events
    rec = dict(E=np.int32(E), seed=np.int64(seed), layouts=five_layouts(),
               ev_type=np.zeros(T, np.int32), ev_arg=np.zeros(T, np.int32), raised=np.zeros(T, np.uint8),
               ball0=np.zeros((T, 2), np.int32), ball1=np.zeros((T, 2), np.int32), goal=np.zeros((T, 2), np.int32),
               fgoal=np.zeros((T, 2), np.int32), fovea0=np.zeros((T, 2), np.int32), fovea1=np.zeros((T, 2), np.int32),
               layout_id=np.zeros(T, np.int32), step_count=np.zeros(T, np.int32),
               foveal_step_count=np.zeros(T, np.int32),
               global_reward=np.zeros(T, np.float64), local_reward=np.zeros(T, np.float64),
               global_done=np.zeros(T, np.uint8), local_done=np.zeros(T, np.uint8),
               visit=np.zeros((T, 18, 18), np.float32),
               fov_planes=np.zeros((T, 7, 5, 5), np.float32), loc_planes=np.zeros((T, 4, 5, 5), np.float32),
               fov_hash=np.zeros(T, np.uint64), loc_hash=np.zeros(T, np.uint64),
               fgoal_plane=np.zeros((T, 5, 5), np.float32))
    state = "reset"
    extra = 0
    for t in range(T):
        fov = loc = None
        try:
            if state == "reset":
                rec["ev_type"][t] = EV_RESET
                fov = env.reset()
                state = "planner" if rs.rand() < 0.9 else "step"     # sometimes step() before any plannerStep
            elif state == "planner":
                rec["ev_type"][t] = EV_PLANNER
                if safe_goals and rs.rand() < 0.7:
                    g = int(env.safeFovealGoal())                    # v6:505-523 (np.random stream)
                else:
                    g = int(rs.randint(0, 25))
                rec["ev_arg"][t] = g
                loc = env.plannerStep(g)
                state = "step"
            else:
                rec["ev_type"][t] = EV_STEP
                a = int(rs.randint(0, 4)) if rs.rand() < 0.93 else int(rs.randint(-1, 7))
                if calm:                                             # back-and-forth: never walks off the 5x5 frame,
                    a = (0, 1, 2, 3)[t % 4]                          # so fovealStepCount reaches its limit of 50
                rec["ev_arg"][t] = a
                with contextlib.redirect_stdout(io.StringIO()):
                    out = env.step(a)
                fov, loc = out[0], out[1]
                assert out[2] == env.globalReward and out[3] == env.originalReward
                assert out[4] == env.globalDone and out[5] == env.localDone and out[7] == a
                assert (out[6] == env.fovealGoal).all() and out[6].shape == (1, 5, 5)
                if env.globalDone:
                    state = "reset"
                elif env.localDone:
                    if extra == 0 and rs.rand() < 0.25:
                        extra = int(rs.randint(1, 3))                # a few more steps with localDone still set
                    if extra > 0:
                        extra -= 1
                        state = "step" if extra > 0 else "planner"
                    else:
                        state = "planner"
        except IndexError:
            rec["raised"][t] = 1                                     # buildLocalObservation indexed outside 5x5
            state = "reset"
        rec["ball0"][t] = (env.ball_x0, env.ball_y0)
        rec["ball1"][t] = (env.ball_x1, env.ball_y1)
        rec["goal"][t] = (env.goal_x, env.goal_y)
        rec["fgoal"][t] = (env.f_goal_x0, env.f_goal_y0)
        rec["fovea0"][t] = (env.fovea_x0, env.fovea_y0)
        rec["fovea1"][t] = (env.fovea_x1, env.fovea_y1)
        rec["layout_id"][t] = layout_id_of(env.grid)
        rec["step_count"][t] = env.stepCount
        rec["foveal_step_count"][t] = env.fovealStepCount
        rec["global_reward"][t] = env.globalReward
        rec["local_reward"][t] = env.originalReward
        rec["global_done"][t] = env.globalDone
        rec["local_done"][t] = env.localDone
        rec["visit"][t] = env.state[2]
        rec["fgoal_plane"][t] = env.fovealGoal[0]
        if fov is not None:
            rec["fov_planes"][t] = unexpand(fov, E)
            rec["fov_hash"][t] = obs_hash(fov)
        if loc is not None:
            rec["loc_planes"][t] = unexpand(loc, E)
            rec["loc_hash"][t] = obs_hash(loc)
    return rec


def save_events(name, rec):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print("wrote %-28s events=%d raised=%d  %.1f KB" % (name, len(rec["ev_type"]), int(rec["raised"].sum()),
                                                        os.path.getsize(path) / 1024.0))


def gen_v5():
    save_events("v5_seed0", rollout_v56("v5", 700, seed=0))
    save_events("v5_seed1", rollout_v56("v5", 500, seed=1))
    save_events("v5_calm_seed3", rollout_v56("v5", 700, seed=3, calm=True))


def gen_v6():
    save_events("v6_seed0", rollout_v56("v6", 600, seed=0, safe_goals=True))
    save_events("v6_seed2", rollout_v56("v6", 400, seed=2, safe_goals=True))


# ----------------------------------------------------------------------------------------
# the placement switches the reference exposes as plain attributes (RANDOM_BALL / RANDOM_GOAL), flipped
# after construction: start cell 'S' (lmaze_env.py:82-89), goal kept from the constructor's reset
# (lmaze_env_v3.py:147) or looked up at 'X' (lmaze_env_v2.py:284-286)
# ----------------------------------------------------------------------------------------
def gen_flags():
    a4 = np.random.RandomState(71).randint(0, 4, 260).astype(np.int32)
    save("v0_fixed_start_seed0", rollout_v0(None, a4, seed=0, random_ball=False))
    save("v3_fixed_start_goal_seed1", rollout_v3(a4, seed=1, random_ball=False, random_goal=False))
    save("v3_fixed_goal_seed2", rollout_v3(a4, seed=2, random_goal=False))
    save("v2_fixed_goal_seed3", rollout_v24("v2", foveal_actions(72, 160), seed=3, random_goal=False))
    save("v2_fixed_start_seed4", rollout_v24("v2", foveal_actions(73, 160), seed=4, random_ball=False))


# ----------------------------------------------------------------------------------------
# 1 000-step rollouts of every step() variant (">= 1 000 random actions incl. out-of-range ids")
# ----------------------------------------------------------------------------------------
def gen_long():
    save("v3_g18_long_seed4", rollout_v3(mixed_actions(81, 1000), seed=4))
    save("v2_long_seed5", rollout_v24("v2", foveal_actions(82, 1000), seed=5))
    save("v4_long_seed6", rollout_v24("v4", foveal_actions(83, 1000), seed=6))
    fg = np.random.RandomState(84).randint(0, 5, (200, 2))
    save("v1_long_seed7", rollout_v1(mixed_actions(85, 1000, lo=-1, hi=5), fg, seed=7))
    save("v0_g12_long_seed8", rollout_v0(None, mixed_actions(86, 1000), seed=8))


# ----------------------------------------------------------------------------------------
# reset() placement distributions: counts per cell over many seeded calls of the reference's own reset()
# (lmaze_env.py:70-78; lmaze_env_v3.py:145-161; lmaze_env_v2.py:90-92,277-299; lmaze_env_v4.py:97-104).  The
# device reset draws from Philox, not from the reference's Mersenne Twister, so it can only be compared in
# distribution -- against THESE counts, not against a list of accepted cells restated by the build.
# expansionRatio is set to 1 on the live object (a plain attribute, as RANDOM_BALL is) so that the x7 loop of
# reset() does not dominate; the placement code draws before it and never reads it.
# ----------------------------------------------------------------------------------------
def gen_reset_hist():
    import contextlib
    import io
    rec = {}
    sink = io.StringIO()

    def fresh(variant, seed):
        five_layouts()
        random.seed(seed)
        np.random.seed(seed)
        env = ref_loader.make(variant)
        env.expansionRatio = 1
        return env

    K0 = 60000
    env = fresh("v0", 101)
    G = env.realgrid
    ball = np.zeros((G, G), np.int64)
    with contextlib.redirect_stdout(sink):
        for _ in range(K0):
            env.reset()
            ball[env.ball_x0, env.ball_y0] += 1
    rec.update(v0_layout=to_codes(env.grid), v0_ball=ball, v0_resets=np.int64(K0))

    env = fresh("v3", 103)
    G = env.realgrid
    goal, ball, same = np.zeros((G, G), np.int64), np.zeros((G, G), np.int64), 0
    with contextlib.redirect_stdout(sink):
        for _ in range(K0):
            env.reset()
            goal[env.goal_x, env.goal_y] += 1
            ball[env.ball_x0, env.ball_y0] += 1
            same += int((env.goal_x, env.goal_y) == (env.ball_x0, env.ball_y0))
    rec.update(v3_layout=to_codes(env.grid), v3_goal=goal, v3_ball=ball, v3_ball_on_goal=np.int64(same), v3_resets=np.int64(K0))

    K2 = 100000
    for variant, seed in (("v2", 102), ("v4", 104)):
        env = fresh(variant, seed)
        G = env.grid.shape[0]
        goal, ball = np.zeros((5, G, G), np.int64), np.zeros((5, G, G), np.int64)
        trans = np.zeros((5, 5), np.int64)
        same = 0
        with contextlib.redirect_stdout(sink):
            for _ in range(K2):
                before = layout_id_of(env.grid)
                env.reset()
                after = layout_id_of(env.grid)
                # v2 places goal and ball on the layout in force BEFORE setGrid (lmaze_env_v2.py:90-92), v4 on the new one
                k = before if variant == "v2" else after
                goal[k, env.goal_x, env.goal_y] += 1
                ball[k, env.ball_x0, env.ball_y0] += 1
                trans[before, after] += 1
                same += int((env.goal_x, env.goal_y) == (env.ball_x0, env.ball_y0))
        rec.update({variant + "_goal": goal, variant + "_ball": ball, variant + "_layout_transitions": trans,
                    variant + "_ball_on_goal": np.int64(same), variant + "_resets": np.int64(K2)})
    rec["layouts"] = five_layouts()
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "reset_hist.npz")
    np.savez_compressed(path, **rec)
    print("wrote reset_hist  %.1f KB" % (os.path.getsize(path) / 1024.0))


GENERATORS = {"reset_hist": gen_reset_hist, "v0": gen_v0, "v3": gen_v3, "v1": gen_v1, "v2": gen_v2, "v4": gen_v4, "v5": gen_v5, "v6": gen_v6,
              "flags": gen_flags, "long": gen_long}

if __name__ == "__main__":
    which = sys.argv[1:] or sorted(GENERATORS)
    for w in which:
        GENERATORS[w]()
