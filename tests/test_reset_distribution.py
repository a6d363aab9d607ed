"""Reset placement against the REFERENCE's own empirical distribution (tests/golden/reset_hist.npz: counts per cell
over 60 000 - 100 000 seeded calls of the reference's reset(), oracle/gen_golden.py gen_reset_hist).  The on-device
reset cannot reproduce the reference's Mersenne Twister stream, so it is compared in distribution: the same support
(exactly the cells the reference ever placed on) and a chi-square homogeneity test of the two count tables.
CPU: the oracle's Philox placement; GPU (-m gpu): the HIP kernels, v0 / v3 / v2 (old-layout quirk) / v4 / v5.
References: lmaze_env.py:70-78, lmaze_env_v3.py:145-161, lmaze_env_v2.py:90-92,277-299, lmaze_env_v4.py:97-104."""
import importlib
import os

import numpy as np
import pytest
from scipy.stats import chi2_contingency

import oracle_lib as O
from helpers import GOLDEN

N = 1 << 18
P_MIN = 1e-4          # seeded inputs: a deterministic verdict, not a flaky one


def hist():
    with np.load(os.path.join(GOLDEN, "reset_hist.npz"), allow_pickle=False) as d:
        return {k: d[k] for k in d.files}


def same_distribution(ours, ref, tag):
    ours, ref = np.asarray(ours, np.int64).ravel(), np.asarray(ref, np.int64).ravel()
    assert ((ours > 0) == (ref > 0)).all(), (tag, "support differs", np.flatnonzero((ours > 0) != (ref > 0)))
    keep = ref > 0
    p = chi2_contingency(np.stack([ours[keep], ref[keep]]))[1]
    assert p > P_MIN, (tag, p)


def counts(xy, G):
    xy = np.asarray(xy)
    return np.bincount(xy[:, 0] * G + xy[:, 1], minlength=G * G)


# ----------------------------------------------------------------------------------------- CPU: the oracle
def test_oracle_v0_and_v3_placement_matches_the_reference_distribution():
    h = hist()
    for variant, vid in (("v0", O.VARIANT_V0), ("v3", O.VARIANT_V3)):
        lay = np.ascontiguousarray(h[variant + "_layout"])
        G = lay.shape[0]
        p = O.params(vid, G)
        ball, goal = np.zeros((N, 2), np.int32), np.zeros((N, 2), np.int32)
        O.reset(p, lay, None, 11, 0, ball, goal if variant == "v3" else None, np.zeros(N, np.int32), np.zeros(N, np.float32),
                np.zeros(N, np.uint8))
        same_distribution(counts(ball, G), h[variant + "_ball"], variant + " ball")
        if variant == "v3":
            same_distribution(counts(goal, G), h["v3_goal"], "v3 goal")
            assert not (ball == goal).all(axis=1).any() and h["v3_ball_on_goal"] == 0


@pytest.mark.parametrize("variant", ["v2", "v4"])
def test_oracle_foveal_placement_matches_the_reference_distribution(variant):
    h = hist()
    layouts = np.ascontiguousarray(h["layouts"])
    G = layouts.shape[-1]
    vid = O.VARIANT_V2 if variant == "v2" else O.VARIANT_V4
    p = O.foveal_params(vid, G, 5)
    n = N // 4
    for k in range(5):
        st = O.FovealState(vid, n, G)
        st.layout_id[:] = k
        O.foveal_reset(p, layouts, None, 1, 13 + k, 0, st)
        if variant == "v2":      # goal and ball were placed on layout k, the one in force BEFORE setGrid
            sel = np.ones(n, bool)
        else:                    # v4: placed on the layout setGrid has just drawn
            sel = st.layout_id == k
        same_distribution(counts(st.goal_xy[sel], G), h[variant + "_goal"][k], (variant, "goal", k))
        same_distribution(counts(st.ball_xy[sel], G), h[variant + "_ball"][k], (variant, "ball", k))
        assert not (st.ball_xy == st.goal_xy).all(axis=1).any()
        same_distribution(np.bincount(st.layout_id, minlength=5), h[variant + "_layout_transitions"].sum(0), (variant, "layout", k))


# ----------------------------------------------------------------------------------------- GPU: the HIP kernels
@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["v0", "v3"])
def test_hip_reset_placement_matches_the_reference_distribution(variant):
    import torch
    pkg = importlib.import_module("gym-lmaze_amd")
    h = hist()
    lay = h[variant + "_layout"]
    G = lay.shape[0]
    env = pkg.LmazeVecEnv(N, variant=variant, layout=lay, seed=29)
    env.reset()                                     # a second epoch, not only the constructor's
    s = env.host_state()
    same_distribution(counts(s["ball_xy"], G), h[variant + "_ball"], variant + " ball")
    if variant == "v3":
        same_distribution(counts(s["goal_xy"], G), h["v3_goal"], "v3 goal")
        assert not (s["ball_xy"] == s["goal_xy"]).all(axis=1).any()
    # the fused auto-reset draws from the same rule: step with every env flagged done
    env.set_state(done=np.ones(N, np.uint8))
    env.step(torch.full((N,), -1, dtype=torch.int32, device=env.device), auto_reset=True)     # id -1: no move
    s = env.host_state()
    same_distribution(counts(s["ball_xy"], G), h[variant + "_ball"], variant + " ball, fused")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["v2", "v4", "v5"])
def test_hip_foveal_reset_placement_matches_the_reference_distribution(variant):
    pkg = importlib.import_module("gym-lmaze_amd")
    h = hist()
    G = h["layouts"].shape[-1]
    ref = "v2" if variant == "v2" else "v4"         # v5/v6 place as v4 does (lmaze_env_v5.py:104-106)
    n = N // 4
    for k in range(5):
        env = pkg.LmazeFovealVecEnv(n, variant=variant, layouts=list(h["layouts"]), seed=31 + k, reset=False)
        env.set_state(layout_id=np.full(n, k, np.int32))
        env.reset()
        s = env.host_state()
        sel = np.ones(n, bool) if variant == "v2" else (s["layout_id"] == k)
        same_distribution(counts(s["goal_xy"][sel], G), h[ref + "_goal"][k], (variant, "goal", k))
        same_distribution(counts(s["ball_xy"][sel], G), h[ref + "_ball"][k], (variant, "ball", k))
        assert not (s["ball_xy"] == s["goal_xy"]).all(axis=1).any()
        same_distribution(np.bincount(s["layout_id"], minlength=5), h[ref + "_layout_transitions"].sum(0), (variant, "layout", k))
