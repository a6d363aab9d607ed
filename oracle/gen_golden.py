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

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


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
def rollout_v0(grid, actions, seed, reset_on_done=True):
    env = ref_loader.make("v0")
    if grid is not None:  # re-size the reference by attribute override
        env.grid = grid
        env.realgrid = grid.shape[0]
        env.gridsize = env.realgrid * env.expansionRatio
    E, G = env.expansionRatio, env.realgrid
    random.seed(seed)
    T = len(actions)
    rec = dict(layout=to_codes(env.grid), E=np.int32(E), seed=np.int64(seed),
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
def rollout_v3(actions, seed, mode="train", grid=None, reset_on_done=True):
    env = ref_loader.make("v3")
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


GENERATORS = {"v0": gen_v0, "v3": gen_v3}

if __name__ == "__main__":
    which = sys.argv[1:] or sorted(GENERATORS)
    for w in which:
        GENERATORS[w]()
