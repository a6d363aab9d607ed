"""TEST INFRASTRUCTURE ONLY (oracle/): numpy/ctypes front end of liblmaze_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this; the
product package never does.  Builds the library with gcc on first use if it is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LMAZE_ORACLE_LIB") or os.path.join(HERE, "liblmaze_oracle.so")   # override: sanitizer build

VARIANT_V0, VARIANT_V3 = 0, 3
LAYOUT_SHARED, LAYOUT_PER_ENV = 0, 1
OBS_BALL, OBS_WALL, OBS_GOAL, OBS_FREE = 1, 2, 4, 8


class Params(C.Structure):
    _fields_ = [("variant", C.c_int32), ("grid", C.c_int32), ("layout_mode", C.c_int32),
                ("step_limit", C.c_int32), ("reward_wall", C.c_float), ("reward_move", C.c_float),
                ("reward_goal", C.c_float), ("launch_hint", C.c_int32)]


def build(force=False):
    if os.environ.get("LMAZE_ORACLE_LIB"):
        return LIB_PATH
    src = os.path.join(HERE, "lmaze_oracle.c")
    hdr = os.path.join(os.path.dirname(HERE), "include", "lmaze.h")
    if (force or not os.path.exists(LIB_PATH)
            or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr),
                                                os.path.getmtime(os.path.join(HERE, "lmaze_oracle_foveal.c")))):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", "liblmaze_oracle.so"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.lmaze_oracle_threads.restype = C.c_int
        # a GPU box shows 256 hardware threads but shares them: stay well inside its process/thread guard
        _lib.lmaze_oracle_set_threads(C.c_int(max(1, min(32, len(os.sched_getaffinity(0))))))
    return _lib


def params(variant, grid, layout_mode=LAYOUT_SHARED, step_limit=100, reward_wall=-1.0,
           reward_move=-0.01, reward_goal=100.0):
    return Params(variant, grid, layout_mode, step_limit, reward_wall, reward_move, reward_goal, 0)


def _p(a, ctype):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(ctype))


def threads():
    return lib().lmaze_oracle_threads()


def set_threads(n):
    lib().lmaze_oracle_set_threads(C.c_int(n))


def step_v0(p, layout, action, ball_xy, step_count, reward, done, goal_count=None, obs=None):
    n = action.shape[0]
    rc = lib().lmaze_oracle_step_v0(C.byref(p), _p(layout, C.c_uint8), _p(action, C.c_int32),
                                    _p(ball_xy, C.c_int32), _p(step_count, C.c_int32),
                                    _p(reward, C.c_float), _p(done, C.c_uint8),
                                    _p(goal_count, C.c_int32), _p(obs, C.c_int32), C.c_int64(n))
    if rc:
        raise RuntimeError("lmaze_oracle_step_v0 -> %d" % rc)


def step_v3(p, layout, action, ball_xy, goal_xy, step_count, reward, done, obs=None):
    n = action.shape[0]
    rc = lib().lmaze_oracle_step_v3(C.byref(p), _p(layout, C.c_uint8), _p(action, C.c_int32),
                                    _p(ball_xy, C.c_int32), _p(goal_xy, C.c_int32),
                                    _p(step_count, C.c_int32), _p(reward, C.c_float),
                                    _p(done, C.c_uint8), _p(obs, C.c_int32), C.c_int64(n))
    if rc:
        raise RuntimeError("lmaze_oracle_step_v3 -> %d" % rc)


def observe(p, layout, ball_xy, goal_xy, obs):
    n = ball_xy.shape[0]
    rc = lib().lmaze_oracle_observe(C.byref(p), _p(layout, C.c_uint8), _p(ball_xy, C.c_int32),
                                    _p(goal_xy, C.c_int32), _p(obs, C.c_int32), C.c_int64(n))
    if rc:
        raise RuntimeError("lmaze_oracle_observe -> %d" % rc)


def reset(p, layout, mask, seed, epoch, ball_xy, goal_xy, step_count, reward, done, obs=None, env_base=0):
    n = ball_xy.shape[0]
    rc = lib().lmaze_oracle_reset(C.byref(p), _p(layout, C.c_uint8), _p(mask, C.c_uint8),
                                  C.c_uint64(seed), C.c_uint64(epoch), C.c_int64(env_base), _p(ball_xy, C.c_int32),
                                  _p(goal_xy, C.c_int32), _p(step_count, C.c_int32),
                                  _p(reward, C.c_float), _p(done, C.c_uint8), _p(obs, C.c_int32),
                                  C.c_int64(n))
    if rc:
        raise RuntimeError("lmaze_oracle_reset -> %d" % rc)


def render_expanded(obs, grid, expansion, channel_mask, out=None):
    n = obs.shape[0]
    cm = np.ascontiguousarray(channel_mask, dtype=np.int32)
    S = grid * expansion
    if out is None:
        out = np.empty((n, len(cm), S, S), dtype=np.float32)
    rc = lib().lmaze_oracle_render_expanded(_p(obs, C.c_int32), C.c_int32(grid), C.c_int32(expansion),
                                            _p(cm, C.c_int32), C.c_int32(len(cm)), _p(out, C.c_float),
                                            C.c_int64(n))
    if rc:
        raise RuntimeError("lmaze_oracle_render_expanded -> %d" % rc)
    return out


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().lmaze_oracle_philox4x32_10(c, k, o)
    return [int(x) for x in o]


# ---------------------------------------------------------------------------------------
# foveal variants (lmaze_oracle_foveal.c)
# ---------------------------------------------------------------------------------------
VARIANT_V1, VARIANT_V2, VARIANT_V4, VARIANT_V5, VARIANT_V6 = 1, 2, 4, 5, 6
FOVEAL_CHANNELS = {VARIANT_V1: 4, VARIANT_V2: 5, VARIANT_V4: 7, VARIANT_V5: 7, VARIANT_V6: 7}


class FovealParams(C.Structure):
    _fields_ = [("variant", C.c_int32), ("grid", C.c_int32), ("n_layouts", C.c_int32), ("step_limit", C.c_int32),
                ("foveal_step_limit", C.c_int32), ("reward_wall", C.c_float), ("reward_move", C.c_float),
                ("reward_goal", C.c_float), ("launch_hint", C.c_int32)]


class FovealBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ball_xy", "goal_xy", "fgoal_xy", "layout_id", "step_count",
                                          "foveal_step_count", "reward", "foveal_reward", "done", "foveal_done",
                                          "visit", "obs", "ball1_xy", "fovea_xy", "last_xy", "foveal_goal",
                                          "obs_local", "visit_clock")]
# `visit` here is the REFERENCE's dense float32[N,G,G] plane, halved whole on every update (lmaze_env_v4.py:211-214);
# the product's clock-relative tiles (include/lmaze.h "The visit map") are compared with it through
# lmaze_foveal_materialise_visit.  visit_clock belongs to that representation: the oracle never reads it (NULL).


def foveal_params(variant, grid, n_layouts):
    if variant == VARIANT_V1:      # lmaze_env_v1.py:22-29
        return FovealParams(variant, grid, n_layouts, 200, 10, -1.0, 0.01, 1.0, 0)
    if variant in (VARIANT_V5, VARIANT_V6):   # lmaze_env_v5.py:45-49
        return FovealParams(variant, grid, n_layouts, 10, 50, -1.0, -0.01, 100.0, 0)
    return FovealParams(variant, grid, n_layouts, 50, 0, -1.0, -0.01, 100.0, 0)   # lmaze_env_v2.py:43-49


class FovealState(object):
    """numpy arrays for one batch + the struct of pointers the C functions take."""

    def __init__(self, variant, n, grid):
        Cn = FOVEAL_CHANNELS[variant]
        self.ball_xy = np.zeros((n, 2), np.int32)
        self.goal_xy = np.zeros((n, 2), np.int32)
        self.fgoal_xy = np.zeros((n, 2), np.int32)
        self.layout_id = np.zeros(n, np.int32)
        self.step_count = np.zeros(n, np.int32)
        self.foveal_step_count = np.zeros(n, np.int32)
        self.reward = np.zeros(n, np.float32)
        self.foveal_reward = np.zeros(n, np.float32)
        self.done = np.zeros(n, np.uint8)
        self.foveal_done = np.zeros(n, np.uint8)
        self.visit = np.zeros((n, grid, grid), np.float32)
        self.obs = np.zeros((n, Cn, 5, 5), np.float32)
        self.ball1_xy = np.zeros((n, 2), np.int32)
        self.fovea_xy = np.zeros((n, 4), np.int32)
        self.last_xy = np.zeros((n, 2), np.int32)
        self.foveal_goal = np.zeros(n, np.int32)
        self.obs_local = np.zeros((n, 4, 5, 5), np.float32)
        self.n = n

    def struct(self):
        return FovealBuffers(*[getattr(self, f[0]).ctypes.data if f[0] != "visit_clock" else None
                               for f in FovealBuffers._fields_])


def foveal_step(p, layouts, action, st):
    b = st.struct()
    rc = lib().lmaze_oracle_foveal_step(C.byref(p), _p(layouts, C.c_uint8), _p(action, C.c_int32), C.byref(b),
                                        C.c_int64(st.n))
    if rc:
        raise RuntimeError("lmaze_oracle_foveal_step -> %d" % rc)


def foveal_reset(p, layouts, mask, place, seed, epoch, st, env_base=0):
    b = st.struct()
    rc = lib().lmaze_oracle_foveal_reset(C.byref(p), _p(layouts, C.c_uint8), _p(mask, C.c_uint8), C.c_int32(place),
                                         C.c_uint64(seed), C.c_uint64(epoch), C.c_int64(env_base), C.byref(b),
                                         C.c_int64(st.n))
    if rc:
        raise RuntimeError("lmaze_oracle_foveal_reset -> %d" % rc)


def v1_set_foveal_goal(p, layouts, ij, mask, st):
    b = st.struct()
    ij = np.ascontiguousarray(ij, dtype=np.int32)
    rc = lib().lmaze_oracle_v1_set_foveal_goal(C.byref(p), _p(layouts, C.c_uint8), _p(ij, C.c_int32),
                                               _p(mask, C.c_uint8), C.byref(b), C.c_int64(st.n))
    if rc:
        raise RuntimeError("lmaze_oracle_v1_set_foveal_goal -> %d" % rc)


def expand_planes(planes, expansion):
    planes = np.ascontiguousarray(planes, dtype=np.float32)
    n, ch, g, _ = planes.shape
    out = np.empty((n, ch, g * expansion, g * expansion), np.float32)
    rc = lib().lmaze_oracle_expand_planes(_p(planes, C.c_float), C.c_int32(ch), C.c_int32(g), C.c_int32(expansion),
                                          _p(out, C.c_float), C.c_int64(n))
    if rc:
        raise RuntimeError("lmaze_oracle_expand_planes -> %d" % rc)
    return out


def v5_step(p, layouts, action, st):
    b = st.struct()
    rc = lib().lmaze_oracle_v5_step(C.byref(p), _p(layouts, C.c_uint8), _p(action, C.c_int32), C.byref(b), C.c_int64(st.n))
    if rc:
        raise RuntimeError("lmaze_oracle_v5_step -> %d" % rc)


def v5_planner_step(p, layouts, goal, mask, st):
    b = st.struct()
    goal = np.ascontiguousarray(goal, dtype=np.int32)
    rc = lib().lmaze_oracle_v5_planner_step(C.byref(p), _p(layouts, C.c_uint8), _p(goal, C.c_int32), _p(mask, C.c_uint8),
                                            C.byref(b), C.c_int64(st.n))
    if rc:
        raise RuntimeError("lmaze_oracle_v5_planner_step -> %d" % rc)


def v5_reset(p, layouts, mask, place, seed, epoch, st, env_base=0):
    b = st.struct()
    rc = lib().lmaze_oracle_v5_reset(C.byref(p), _p(layouts, C.c_uint8), _p(mask, C.c_uint8), C.c_int32(place),
                                     C.c_uint64(seed), C.c_uint64(epoch), C.c_int64(env_base), C.byref(b), C.c_int64(st.n))
    if rc:
        raise RuntimeError("lmaze_oracle_v5_reset -> %d" % rc)


def v5_hier_step(p, layouts, action, goal, seed, epoch, st, env_base=0):
    """reset where done, plannerStep(goal) where localDone or just reset, step(action): the composition the
    fused HIP launch lmaze_v5_hier_step is checked against."""
    b = st.struct()
    goal = np.ascontiguousarray(goal, dtype=np.int32)
    rc = lib().lmaze_oracle_v5_hier_step(C.byref(p), _p(layouts, C.c_uint8), _p(action, C.c_int32), _p(goal, C.c_int32),
                                         C.byref(b), C.c_int64(st.n), C.c_uint64(seed), C.c_uint64(epoch), C.c_int64(env_base))
    if rc:
        raise RuntimeError("lmaze_oracle_v5_hier_step -> %d" % rc)


def v6_safe_foveal_goal(p, layouts, seed, epoch, st, env_base=0):
    b = st.struct()
    out = np.zeros(st.n, np.int32)
    rc = lib().lmaze_oracle_v6_safe_foveal_goal(C.byref(p), _p(layouts, C.c_uint8), C.c_uint64(seed), C.c_uint64(epoch),
                                                C.c_int64(env_base), C.byref(b), _p(out, C.c_int32), C.c_int64(st.n))
    if rc:
        raise RuntimeError("lmaze_oracle_v6_safe_foveal_goal -> %d" % rc)
    return out
