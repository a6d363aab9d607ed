"""CPU: the C oracle (oracle/lmaze_oracle.c) replays every golden fixture bit-exactly.

The fixtures were produced by the reference's own step()/reset() (oracle/gen_golden.py);
this is what pins the oracle ("parity pinned" in its header)."""
import numpy as np
import pytest

import oracle_lib as O
from conftest import golden_files
from helpers import (V0_CHANNEL_MASK, V3_CHANNEL_MASK, compact_to_ref_bits, f32_bits, load_golden,
                     obs_hash, ref_reward_bits)


def _replay(g, variant):
    G = g["layout"].shape[0]
    E = int(g["E"])
    T = len(g["actions"])
    layout = np.ascontiguousarray(g["layout"])
    v3 = variant == O.VARIANT_V3
    p = O.params(variant, G)
    cmask = V3_CHANNEL_MASK if v3 else V0_CHANNEL_MASK
    ball = np.zeros((1, 2), np.int32)
    goal = np.zeros((1, 2), np.int32)
    sc = np.zeros(1, np.int32)
    rew = np.zeros(1, np.float32)
    done = np.zeros(1, np.uint8)
    gc = np.zeros(1, np.int32)
    obs = np.zeros((1, G, G), np.int32)
    n_reset = 0
    for t in range(T):
        if g["reset_before"][t]:
            # inject the reference's own placement (its RNG stream is not reproduced on device)
            ball[0] = g["ball_before"][t]
            if v3:
                goal[0] = g["goal_before"][t]
            sc[0] = 0
            rew[0] = -0.0
            O.observe(p, layout, ball, goal if v3 else None, obs)
            assert (compact_to_ref_bits(obs[0], cmask) == g["reset_planes"][n_reset]).all()
            full = O.render_expanded(obs, G, E, cmask)
            assert obs_hash(full[0]) == g["reset_hash"][n_reset]
            n_reset += 1
        assert tuple(ball[0]) == tuple(g["ball_before"][t])
        a = g["actions"][t:t + 1].copy()
        if v3:
            O.step_v3(p, layout, a, ball, goal, sc, rew, done, obs)
        else:
            O.step_v0(p, layout, a, ball, sc, rew, done, gc, obs)
        assert f32_bits(rew)[0] == ref_reward_bits(g["reward"][t]), (t, rew[0], g["reward"][t])
        assert done[0] == g["done"][t], t
        assert tuple(ball[0]) == tuple(g["ball"][t]), t
        assert sc[0] == g["step_count"][t], t
        if not v3:
            assert gc[0] == g["goal_count"][t], t
        assert (compact_to_ref_bits(obs[0], cmask) == g["planes"][t]).all(), t
        full = O.render_expanded(obs, G, E, cmask)
        assert obs_hash(full[0]) == g["obs_hash"][t], t
    assert n_reset == len(g["reset_hash"])


@pytest.mark.parametrize("name", golden_files("v0_"))
def test_oracle_v0_matches_reference(name):
    _replay(load_golden(name), O.VARIANT_V0)


@pytest.mark.parametrize("name", golden_files("v3_"))
def test_oracle_v3_matches_reference(name):
    _replay(load_golden(name), O.VARIANT_V3)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert O.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                           [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


# ---------------------------------------------------------------------------------------
# foveal variants: v1 (setFovealGoal + two reward streams), v2 (teleport), v4 (visit map)
# ---------------------------------------------------------------------------------------
def _replay_v24(g, variant):
    layouts = np.ascontiguousarray(g["layouts"])
    G, E = layouts.shape[-1], int(g["E"])
    p = O.foveal_params(variant, G, layouts.shape[0])
    st = O.FovealState(variant, 1, G)
    n_reset = 0
    for t in range(len(g["actions"])):
        if g["reset_before"][t]:
            st.ball_xy[0] = g["ball_before"][t]
            st.goal_xy[0] = g["goal_before"][t]
            st.layout_id[0] = g["layout_id"][t]
            O.foveal_reset(p, layouts, None, 0, 0, 0, st)           # place=0: the reference's own draws
            assert (st.obs[0].view(np.uint32) == g["reset_planes"][n_reset].view(np.uint32)).all()
            assert obs_hash(O.expand_planes(st.obs, E)[0]) == g["reset_hash"][n_reset]
            n_reset += 1
        assert tuple(st.ball_xy[0]) == tuple(g["ball_before"][t]) and st.layout_id[0] == g["layout_id"][t]
        O.foveal_step(p, layouts, g["actions"][t:t + 1].copy(), st)
        assert f32_bits(st.reward)[0] == ref_reward_bits(g["reward"][t]), t
        assert st.done[0] == g["done"][t] and st.step_count[0] == g["step_count"][t], t
        assert tuple(st.ball_xy[0]) == tuple(g["ball"][t]), t
        assert (st.obs[0].view(np.uint32) == g["planes"][t].view(np.uint32)).all(), t
        assert obs_hash(O.expand_planes(st.obs, E)[0]) == g["obs_hash"][t], t
        if "visit" in g:        # v4_deepdecay: the whole plane, incl. cells decayed below 2^-126 (subnormal, rounded every step)
            assert (st.visit[0].view(np.uint32) == g["visit"][t].view(np.uint32)).all(), t
    assert n_reset == len(g["reset_hash"])


@pytest.mark.parametrize("name", golden_files("v2_"))
def test_oracle_v2_matches_reference(name):
    _replay_v24(load_golden(name), O.VARIANT_V2)


@pytest.mark.parametrize("name", golden_files("v4_"))
def test_oracle_v4_matches_reference(name):
    _replay_v24(load_golden(name), O.VARIANT_V4)


@pytest.mark.parametrize("name", golden_files("v1_"))
def test_oracle_v1_matches_reference(name):
    g = load_golden(name)
    layout = np.ascontiguousarray(g["layout"])[None]
    G, E = layout.shape[-1], int(g["E"])
    p = O.foveal_params(O.VARIANT_V1, G, 1)
    st = O.FovealState(O.VARIANT_V1, 1, G)
    n_reset = 0
    for t in range(len(g["actions"])):
        if g["reset_before"][t]:
            O.foveal_reset(p, layout, None, 1, 0, 0, st)            # v1 placement is deterministic ('S')
            assert (st.obs[0].view(np.uint32) == g["reset_planes"][n_reset].view(np.uint32)).all()
            assert obs_hash(O.expand_planes(st.obs, E)[0]) == g["reset_hash"][n_reset]
            n_reset += 1
        if g["setgoal_before"][t]:
            O.v1_set_foveal_goal(p, layout, g["setgoal_ij"][t:t + 1], None, st)
            assert (st.obs[0].view(np.uint32) == g["setgoal_planes"][t].view(np.uint32)).all(), t
        assert tuple(st.ball_xy[0]) == tuple(g["ball_before"][t]) and tuple(st.fgoal_xy[0]) == tuple(g["fgoal_before"][t])
        assert st.foveal_step_count[0] == g["fstep_before"][t]
        O.foveal_step(p, layout, g["actions"][t:t + 1].copy(), st)
        assert f32_bits(st.reward)[0] == ref_reward_bits(g["reward"][t]), t
        assert f32_bits(st.foveal_reward)[0] == ref_reward_bits(g["foveal_reward"][t]), t
        assert st.done[0] == g["done"][t] and st.foveal_done[0] == g["foveal_done"][t], t
        assert st.step_count[0] == g["step_count"][t] and st.foveal_step_count[0] == g["foveal_step_count"][t], t
        assert tuple(st.ball_xy[0]) == tuple(g["ball"][t]), t
        assert (st.obs[0].view(np.uint32) == g["planes"][t].view(np.uint32)).all(), t
        assert obs_hash(O.expand_planes(st.obs, E)[0]) == g["obs_hash"][t], t
    assert n_reset == len(g["reset_hash"])


# ---------------------------------------------------------------------------------------
# v5 / v6: event rollouts (reset / plannerStep / step), full state compared after every event
# ---------------------------------------------------------------------------------------
def check_v56_state(st, g, t, i=0):
    assert tuple(st.ball_xy[i]) == tuple(g["ball0"][t]) and tuple(st.ball1_xy[i]) == tuple(g["ball1"][t]), t
    assert tuple(st.goal_xy[i]) == tuple(g["goal"][t]) and tuple(st.fgoal_xy[i]) == tuple(g["fgoal"][t]), t
    assert tuple(st.fovea_xy[i]) == tuple(g["fovea0"][t]) + tuple(g["fovea1"][t]), t
    assert st.layout_id[i] == g["layout_id"][t], t
    assert st.step_count[i] == g["step_count"][t] and st.foveal_step_count[i] == g["foveal_step_count"][t], t
    assert f32_bits(st.reward)[i] == ref_reward_bits(g["global_reward"][t]), t
    assert f32_bits(st.foveal_reward)[i] == ref_reward_bits(g["local_reward"][t]), t
    assert st.done[i] == g["global_done"][t] and st.foveal_done[i] == g["local_done"][t], t
    assert (st.visit[i].view(np.uint32) == g["visit"][t].view(np.uint32)).all(), t
    plane = np.zeros(25, np.float32)
    plane[st.foveal_goal[i]] = 1.0
    assert (plane.reshape(5, 5) == g["fgoal_plane"][t]).all(), t


@pytest.mark.parametrize("name", golden_files("v5_") + golden_files("v6_"))
def test_oracle_v56_matches_reference(name):
    g = load_golden(name)
    layouts = np.ascontiguousarray(g["layouts"])
    G, E = layouts.shape[-1], int(g["E"])
    p = O.foveal_params(O.VARIANT_V5, G, layouts.shape[0])
    st = O.FovealState(O.VARIANT_V5, 1, G)
    for t in range(len(g["ev_type"])):
        ev, arg = int(g["ev_type"][t]), int(g["ev_arg"][t])
        if ev == 0:
            st.ball_xy[0] = g["ball0"][t]
            st.goal_xy[0] = g["goal"][t]
            st.layout_id[0] = g["layout_id"][t]
            O.v5_reset(p, layouts, None, 0, 0, 0, st)
        elif ev == 1:
            O.v5_planner_step(p, layouts, np.array([arg], np.int32), None, st)
        else:
            O.v5_step(p, layouts, np.array([arg], np.int32), st)
        check_v56_state(st, g, t)
        if g["raised"][t]:
            continue            # the reference raised inside buildLocalObservation: no observation to compare
        if ev in (0, 2):
            assert (st.obs[0].view(np.uint32) == g["fov_planes"][t].view(np.uint32)).all(), t
            assert obs_hash(O.expand_planes(st.obs, E)[0]) == g["fov_hash"][t], t
        if ev in (1, 2):
            assert (st.obs_local[0].view(np.uint32) == g["loc_planes"][t].view(np.uint32)).all(), t
            assert obs_hash(O.expand_planes(st.obs_local, E)[0]) == g["loc_hash"][t], t


# ---------------------------------------------------------------------------------------
# the NumPy-vectorised restatement (oracle/oracle_numpy.py, a cpu_baseline leg of bench.py)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_files("v0_"))
def test_numpy_v0_matches_reference(name):
    import oracle_numpy as ON
    g = load_golden(name)
    layout = np.ascontiguousarray(g["layout"])
    G = layout.shape[0]
    static = ON.static_bits(layout)
    ball = np.zeros((1, 2), np.int32)
    sc, gc = np.zeros(1, np.int32), np.zeros(1, np.int32)
    rew, done = np.zeros(1, np.float32), np.zeros(1, np.uint8)
    obs = np.zeros((1, G, G), np.int32)
    for t in range(len(g["actions"])):
        if g["reset_before"][t]:
            ball[0] = g["ball_before"][t]
            sc[0] = 0
            rew[0] = -0.0
        ON.step_v0(layout, static, g["actions"][t:t + 1].astype(np.int32), ball, sc, rew, done, gc, obs)
        assert f32_bits(rew)[0] == ref_reward_bits(g["reward"][t]), t
        assert done[0] == g["done"][t] and tuple(ball[0]) == tuple(g["ball"][t]) and sc[0] == g["step_count"][t], t
        assert gc[0] == g["goal_count"][t], t
        assert (compact_to_ref_bits(obs[0], V0_CHANNEL_MASK) == g["planes"][t]).all(), t


@pytest.mark.parametrize("per_env", [False, True])
def test_numpy_v0_matches_c_oracle_batched(per_env):
    import oracle_numpy as ON
    from helpers import bordered_random_layouts, random_free_cells
    rs = np.random.RandomState(11)
    n, G = 3000, 11
    lays = bordered_random_layouts(n if per_env else 1, G, 5)
    lay = lays if per_env else lays[0]
    p = O.params(O.VARIANT_V0, G, O.LAYOUT_PER_ENV if per_env else O.LAYOUT_SHARED)
    ball = random_free_cells(lays if per_env else np.repeat(lays, n, 0), 6)
    st = [dict(ball=ball.copy(), sc=np.zeros(n, np.int32), rew=np.zeros(n, np.float32), done=np.zeros(n, np.uint8),
               gc=np.zeros(n, np.int32), obs=np.zeros((n, G, G), np.int32)) for _ in range(2)]
    static = ON.static_bits(lay)
    for t in range(150):
        a = rs.randint(-1, 6, n).astype(np.int32)
        O.step_v0(p, np.ascontiguousarray(lay), a, st[0]["ball"], st[0]["sc"], st[0]["rew"], st[0]["done"], st[0]["gc"], st[0]["obs"])
        ON.step_v0(lay, static, a, st[1]["ball"], st[1]["sc"], st[1]["rew"], st[1]["done"], st[1]["gc"], st[1]["obs"])
        for k in ("ball", "sc", "done", "gc", "obs"):
            assert (st[0][k] == st[1][k]).all(), (t, k)
        assert (f32_bits(st[0]["rew"]) == f32_bits(st[1]["rew"])).all(), t
