"""TEST INFRASTRUCTURE ONLY (oracle/): numpy/ctypes front end of liblmaze_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this; the
product package never does.  Builds the library with gcc on first use if it is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liblmaze_oracle.so")

VARIANT_V0, VARIANT_V3 = 0, 3
LAYOUT_SHARED, LAYOUT_PER_ENV = 0, 1
OBS_BALL, OBS_WALL, OBS_GOAL, OBS_FREE = 1, 2, 4, 8


class Params(C.Structure):
    _fields_ = [("variant", C.c_int32), ("grid", C.c_int32), ("layout_mode", C.c_int32),
                ("step_limit", C.c_int32), ("reward_wall", C.c_float), ("reward_move", C.c_float),
                ("reward_goal", C.c_float), ("launch_hint", C.c_int32)]


def build(force=False):
    src = os.path.join(HERE, "lmaze_oracle.c")
    hdr = os.path.join(os.path.dirname(HERE), "include", "lmaze.h")
    if (force or not os.path.exists(LIB_PATH)
            or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", "liblmaze_oracle.so"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.lmaze_oracle_threads.restype = C.c_int
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
