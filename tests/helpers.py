"""Shared helpers for the parity tests (fixture replay, hashing, layouts)."""
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as d:
        return {k: d[k] for k in d.files}


def obs_hash(obs):
    obs = np.ascontiguousarray(obs, dtype=np.float32)
    return np.frombuffer(hashlib.sha256(obs.tobytes()).digest()[:8], dtype="<u8")[0]


def f32_bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def ref_reward_bits(r64):
    """The reference returns Python doubles; the build keeps float32.  Compare bit patterns
    of float32(reference) -- this keeps the sign of the reference's literal -0.0."""
    return np.asarray(r64, dtype=np.float64).astype(np.float32).view(np.uint32)


# reference channel order -> LMAZE_OBS_* bit of the compact plane (include/lmaze.h)
V0_CHANNEL_MASK = (1, 2, 4, 8)   # ball, wall, goal, blank   (lmaze_env.py:208-215)
V3_CHANNEL_MASK = (8, 1, 4)      # free, ball, goal          (lmaze_env_v3.py:291-293)


def compact_to_ref_bits(obs, channel_mask):
    """int32 compact planes [..,G,G] -> uint8, bit c = reference channel c (fixture packing)."""
    out = np.zeros(obs.shape, dtype=np.uint8)
    for c, m in enumerate(channel_mask):
        out |= ((obs & m) != 0).astype(np.uint8) << c
    return out


def bordered_random_layouts(n, G, seed, p_wall=0.25):
    """uint8[n,G,G] random mazes: 'W' border, interior walls with prob p_wall, one 'X', one 'S'."""
    rs = np.random.RandomState(seed)
    lay = np.where(rs.rand(n, G, G) < p_wall, ord("W"), ord("B")).astype(np.uint8)
    lay[:, 0, :] = lay[:, -1, :] = lay[:, :, 0] = lay[:, :, -1] = ord("W")
    # guarantee two free interior cells, then mark X and S
    lay[:, 1, 1] = ord("B")
    lay[:, G - 2, G - 2] = ord("B")
    flat = lay.reshape(n, -1)
    for i in range(n):
        free = np.flatnonzero(flat[i] == ord("B"))
        x, s = rs.choice(free, 2, replace=False)
        flat[i, x] = ord("X")
        flat[i, s] = ord("S")
    return lay


def random_free_cells(layouts, seed, forbid=(ord("W"), ord("X"))):
    """one uniformly chosen allowed cell per env -> int32[n,2] (x,y)."""
    rs = np.random.RandomState(seed)
    n, G, _ = layouts.shape
    out = np.zeros((n, 2), np.int32)
    flat = layouts.reshape(n, -1)
    for i in range(n):
        ok = np.flatnonzero(~np.isin(flat[i], forbid))
        c = ok[rs.randint(len(ok))]
        out[i] = (c // G, c % G)
    return out
