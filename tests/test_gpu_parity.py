"""GPU: the HIP step path (through the C ABI) against the golden fixtures and the C oracle.

Bit-exact everywhere: integers, bytes, and float32 bit patterns of reward (incl. -0.0)."""
import importlib
import random

import numpy as np
import pytest
import torch

import oracle_lib as O
from conftest import golden_files
from helpers import (V0_CHANNEL_MASK, V3_CHANNEL_MASK, bordered_random_layouts, compact_to_ref_bits, f32_bits,
                     load_golden, obs_hash, random_free_cells, ref_reward_bits)

pytestmark = pytest.mark.gpu

PKG = importlib.import_module("gym-lmaze_amd")
L = PKG.layouts


def _np(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------------------
# 1. golden fixtures through the batched engine (N = 1, placement injected from the fixture)
# ----------------------------------------------------------------------------------------
def _replay_core(g, variant):
    v3 = variant == "v3"
    cmask = V3_CHANNEL_MASK if v3 else V0_CHANNEL_MASK
    E = int(g["E"])
    env = PKG.LmazeVecEnv(1, variant=variant, layout=g["layout"], expansion=E)
    T = len(g["actions"])
    n_reset = 0
    for t in range(T):
        if g["reset_before"][t]:
            env.set_state(ball_xy=g["ball_before"][t:t + 1], step_count=np.zeros(1, np.int32),
                          reward=np.array([-0.0], np.float32), done=np.zeros(1, np.uint8),
                          goal_xy=g["goal_before"][t:t + 1] if v3 else None)
            obs = _np(env.observe())
            assert (compact_to_ref_bits(obs[0], cmask) == g["reset_planes"][n_reset]).all()
            assert obs_hash(_np(env.expanded())[0]) == g["reset_hash"][n_reset]
            n_reset += 1
        obs, rew, done, _ = env.step(g["actions"][t:t + 1])
        h = env.host_state()
        assert f32_bits(h["reward"])[0] == ref_reward_bits(g["reward"][t]), t
        assert h["done"][0] == g["done"][t], t
        assert tuple(h["ball_xy"][0]) == tuple(g["ball"][t]), t
        assert h["step_count"][0] == g["step_count"][t], t
        if not v3:
            assert h["goal_count"][0] == g["goal_count"][t], t   # persists across resets (lmaze_env.py:24)
        assert (compact_to_ref_bits(_np(obs)[0], cmask) == g["planes"][t]).all(), t
        assert obs_hash(_np(env.expanded())[0]) == g["obs_hash"][t], t


@pytest.mark.parametrize("name", golden_files("v0_"))
def test_hip_v0_matches_reference_fixture(name):
    _replay_core(load_golden(name), "v0")


@pytest.mark.parametrize("name", golden_files("v3_"))
def test_hip_v3_matches_reference_fixture(name):
    _replay_core(load_golden(name), "v3")


# ----------------------------------------------------------------------------------------
# 2. the drop-in classes end to end (BASELINE configs[0]): reference-typed returns, and the
#    host reset consuming `random` in the reference's draw order reproduces its start cells
# ----------------------------------------------------------------------------------------
def test_dropin_v0_c1_rollout(capsys):
    g = load_golden("v0_c1_g12_seed0")
    import gym_lmaze
    env = gym_lmaze.make("lmaze-v0")
    assert "init-init" in capsys.readouterr().out
    assert env.observation_space.shape == (4, 84, 84) and env.action_space.n == 4
    random.seed(int(g["seed"]))
    n_reset = 0
    need_reset = True
    for t in range(len(g["actions"])):
        if need_reset:
            o = env.reset()
            assert isinstance(o, np.ndarray) and o.dtype == np.float32 and o.shape == (4, 84, 84)
            assert obs_hash(o) == g["reset_hash"][n_reset]
            assert (env.ball_x0, env.ball_y0) == tuple(g["ball_before"][t])
            n_reset += 1
            need_reset = False
        a = int(g["actions"][t])
        o, r, d, info = env.step(a)
        assert type(r) is float and type(d) is bool and info == a and isinstance(o, np.ndarray)
        assert r == g["reward"][t] and d == bool(g["done"][t]), t   # exact doubles, e.g. -0.01
        assert obs_hash(o) == g["obs_hash"][t], t
        assert (env.ball_x0, env.ball_y0) == tuple(g["ball"][t])
        assert env.stepCount == g["step_count"][t]
        need_reset = d
    assert (env.goal_x, env.goal_y) == tuple(g["goal"])
    assert env.state.shape == (4 * 12 * 12,) and env.state.dtype == np.float32


def test_dropin_v0_resize_by_attribute_override():
    """the reference is re-sized by overriding grid/realgrid/gridsize, then reset()."""
    g = load_golden("v0_g11_open_seed0")
    from gym_lmaze.envs import LmazeEnv
    env = LmazeEnv()
    env.grid = np.vectorize(chr)(g["layout"])
    env.realgrid = 11
    env.gridsize = 77
    random.seed(int(g["seed"]))
    need_reset, n_reset = True, 0
    for t in range(150):
        if need_reset:
            o = env.reset()
            assert o.shape == (4, 77, 77) and obs_hash(o) == g["reset_hash"][n_reset]
            n_reset += 1
            need_reset = False
        o, r, d, _ = env.step(int(g["actions"][t]))
        assert obs_hash(o) == g["obs_hash"][t] and np.float32(r).view(np.uint32) == ref_reward_bits(g["reward"][t])
        need_reset = d


def test_dropin_v3_strings_and_shared_buffer():
    g = load_golden("v3_g18_seed0")
    import gym_lmaze
    env = gym_lmaze.make("lmaze-v3")
    assert env.observation_space.shape == (3, 72, 72)
    random.seed(int(g["seed"]))
    need_reset, n_reset, first = True, 0, None
    for t in range(len(g["actions"])):
        if need_reset:
            o = env.reset()
            assert obs_hash(o) == g["reset_hash"][n_reset]
            assert (env.goal_x, env.goal_y) == tuple(g["goal_before"][t])
            n_reset += 1
            need_reset = False
        a = int(g["actions"][t])
        arg = str(a) if 0 <= a <= 3 else a
        o, r, d, info = env.step(arg)
        first = o if first is None else first
        assert o is first                      # one reused buffer (lmaze_env_v3.py:122,400)
        assert info is arg or info == arg
        assert r == g["reward"][t] and d == bool(g["done"][t]), t
        assert obs_hash(o) == g["obs_hash"][t], t
        need_reset = d
    # named strings move too; an int never does (lmaze_env_v3.py:236-247)
    env.reset(mode="test")
    assert (env.ball_x0, env.ball_y0, env.goal_x, env.goal_y) == (7, 8, 8, 8)
    env.step("left")
    assert (env.ball_x0, env.ball_y0) == (6, 8)
    env.step(1)
    assert (env.ball_x0, env.ball_y0) == (6, 8)


# ----------------------------------------------------------------------------------------
# 3. batched engine vs the C oracle on identical seeded inputs
# ----------------------------------------------------------------------------------------
def _oracle_state(env, variant, layout_np):
    h = env.host_state()
    st = {k: np.array(v, copy=True) for k, v in h.items()}
    p = O.params(O.VARIANT_V3 if variant == "v3" else O.VARIANT_V0, env.grid,
                 O.LAYOUT_PER_ENV if layout_np.ndim == 3 else O.LAYOUT_SHARED, env.step_limit, *env.rewards)
    return p, st


def _compare_rollout(variant, N, G, T, shared=True, seed=0, render_every=1, action_hi=5, layout=None):
    rs = np.random.RandomState(seed)
    if shared:
        lay = layout if layout is not None else bordered_random_layouts(1, G, seed + 100)[0]
        env = PKG.LmazeVecEnv(N, variant=variant, layout=lay, seed=seed)
        lay_all = np.broadcast_to(lay, (N, G, G))
    else:
        lay = bordered_random_layouts(N, G, seed + 100)
        env = PKG.LmazeVecEnv(N, variant=variant, per_env_layouts=lay, seed=seed)
        lay_all = lay
    ball = random_free_cells(lay_all, seed + 1, forbid=(ord("W"),) if variant == "v3" else (ord("W"), ord("X")))
    env.set_state(ball_xy=ball)
    if variant == "v3":
        env.set_state(goal_xy=random_free_cells(lay_all, seed + 2, forbid=(ord("W"),)))
    p, st = _oracle_state(env, variant, lay)
    lay_c = np.ascontiguousarray(lay)
    obs_ref = np.zeros((N, G, G), np.int32)
    for t in range(T):
        # mostly legal ids, some out-of-range ones (-1 .. action_hi)
        a = np.where(rs.rand(N) < 0.85, rs.randint(0, 4, N), rs.randint(-1, action_hi + 1, N)).astype(np.int32)
        obs, rew, done, _ = env.step(torch.from_numpy(a))
        if variant == "v3":
            O.step_v3(p, lay_c, a, st["ball_xy"], st["goal_xy"], st["step_count"], st["reward"], st["done"], obs_ref)
        else:
            O.step_v0(p, lay_c, a, st["ball_xy"], st["step_count"], st["reward"], st["done"], st["goal_count"],
                      obs_ref)
        h = env.host_state()
        for k in ("ball_xy", "step_count", "goal_count", "done"):
            assert (h[k] == st[k]).all(), (k, t)
        assert (f32_bits(h["reward"]) == f32_bits(st["reward"])).all(), t
        if t % render_every == 0 or t == T - 1:
            assert (_np(obs) == obs_ref).all(), t
    return env, st


def test_c2_65536_8x8_every_step_bit_exact():
    """BASELINE configs[1]: 65 536 parallel 8x8 mazes, int32 state, bit-exact vs the CPU step()."""
    _compare_rollout("v0", 65536, 8, 256, shared=True, seed=1, layout=L.to_codes(L.GRID_8_BORDERED))


@pytest.mark.parametrize("G", [8, 11, 12, 14, 18, 32])
@pytest.mark.parametrize("variant", ["v0", "v3"])
def test_shared_layout_specialised_grids(variant, G):
    # N not a multiple of 256 nor of 4: exercises the partial last workgroup and ragged tail
    _compare_rollout(variant, 3003, G, 40, shared=True, seed=G)


@pytest.mark.parametrize("G", [4, 5, 9, 10, 13, 21, 33, 64])
@pytest.mark.parametrize("variant", ["v0", "v3"])
def test_shared_layout_generic_grids(variant, G):
    _compare_rollout(variant, 1030 if G < 40 else 300, G, 24, shared=True, seed=G)


@pytest.mark.parametrize("G", [8, 11, 12, 32, 7, 9, 21, 40, 16, 48, 64])
@pytest.mark.parametrize("variant", ["v0", "v3"])
def test_per_env_layouts(variant, G):
    _compare_rollout(variant, 1537 if G < 40 else 200, G, 24, shared=False, seed=G + 7)


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 63, 64, 255, 256, 257, 511])
def test_small_and_ragged_batches(N):
    _compare_rollout("v0", N, 11, 12, shared=True, seed=N)
    _compare_rollout("v0", N, 11, 8, shared=False, seed=N)


def test_planes_are_the_bit_tests_of_the_compact_obs():
    env = PKG.LmazeVecEnv(500, variant="v3", seed=2)
    env.step(torch.randint(0, 4, (500,), dtype=torch.int32))
    p = env.planes()
    o = env.obs
    want = torch.stack([((o & m) != 0).float() for m in env.channel_mask], dim=1)
    assert p.shape == (500, 3, 18, 18) and (p == want).all()


def test_transition_only_leaves_obs_untouched():
    env = PKG.LmazeVecEnv(1000, variant="v0", layout=L.open_room(11, (5, 5)))
    before = env.obs.clone()
    env.step(torch.randint(0, 4, (1000,), dtype=torch.int32), render=False)
    assert (env.obs == before).all()
    assert (env.step_count == 1).all()


def test_sticky_reward_and_done_quirks_known_answers():
    """Appendix B-1/2/3 on the 12x12 layout: sticky reward on 'S', in-place move on unknown id,
    done only at stepCount == 100."""
    env = PKG.LmazeVecEnv(1, variant="v0")
    env.set_state(ball_xy=[[1, 2]], step_count=[0], reward=[-0.0])
    env.step([0])                              # up into the border wall
    assert float(env.reward[0]) == -1.0
    env.step([2])                              # left into 'S' at (1,1): no branch fires
    h = env.host_state()
    assert tuple(h["ball_xy"][0]) == (1, 2) and h["reward"][0] == -1.0
    env.step([7])                              # unknown id on a 'B' cell: moves in place, -0.01
    h = env.host_state()
    assert tuple(h["ball_xy"][0]) == (1, 2) and h["reward"][0] == np.float32(-0.01)
    assert int(_np(env.obs)[0].__and__(1).sum()) == 1
    env.set_state(step_count=[98])
    assert not bool(env.step([7])[2][0])       # 99
    env.set_state(step_count=[99])
    assert bool(env.step([7])[2][0])           # stepCount == 100
    assert not bool(env.step([7])[2][0])       # 101: False again
    # goal: from (5,6)? 'X' is at (5,5); (4,5) is 'B' above it
    env.set_state(ball_xy=[[4, 5]], step_count=[0], goal_count=[0])
    _, r, d, _ = env.step([1])
    assert float(r[0]) == 100.0 and bool(d[0]) and int(env.goal_count[0]) == 1
    _, r, d, _ = env.step([9])                 # unknown id standing on 'X': scores again
    assert float(r[0]) == 100.0 and int(env.goal_count[0]) == 2


def test_v3_lookahead_known_answers():
    """Appendix A v3: two cells short and stepping toward the goal scores; stepping onto it does not."""
    env = PKG.LmazeVecEnv(1, variant="v3")
    env.set_state(ball_xy=[[6, 8]], goal_xy=[[8, 8]], step_count=[0])
    _, r, d, _ = env.step([1])                 # (6,8)->(7,8); look-ahead (8,8) == goal
    assert float(r[0]) == 100.0 and bool(d[0])
    _, r, d, _ = env.step([1])                 # onto the goal: look-ahead (9,8) != goal
    assert float(r[0]) == np.float32(-0.01) and not bool(d[0])
    _, r, d, _ = env.step([-1])                # no-op standing on the goal
    assert float(r[0]) == 100.0
    env.set_state(step_count=[99])
    assert not bool(env.step([0])[2][0])       # 100 > 100 False
    assert bool(env.step([0])[2][0])           # 101 > 100


# ----------------------------------------------------------------------------------------
# 4. reset kernel vs oracle (same Philox draws), and its distribution
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["v0", "v3"])
@pytest.mark.parametrize("shared,G", [(True, 12), (False, 12), (False, 16), (False, 32), (False, 48), (False, 64)])
def test_reset_matches_oracle(variant, shared, G):
    N, seed = 5000 if G < 40 else 700, 1234
    if shared:
        lay = L.to_codes(L.V0_GRID_12)
        env = PKG.LmazeVecEnv(N, variant=variant, layout=lay, seed=seed, env_base=77)
    else:
        lay = bordered_random_layouts(N, G, 5)
        env = PKG.LmazeVecEnv(N, variant=variant, per_env_layouts=lay, seed=seed, env_base=77)
    p, st = _oracle_state(env, variant, lay)
    # constructor already ran epoch 0; replay it in the oracle from a zero state
    for k in st:
        st[k][...] = 0
    st["goal_xy"][...] = _np(env.goal_xy) if variant == "v0" else 0
    obs_ref = np.zeros((N, G, G), np.int32)
    O.reset(p, np.ascontiguousarray(lay), None, seed, 0, st["ball_xy"], st["goal_xy"] if variant == "v3" else None,
            st["step_count"], st["reward"], st["done"], obs_ref, env_base=77)
    h = env.host_state()
    assert (h["ball_xy"] == st["ball_xy"]).all()
    if variant == "v3":
        assert (h["goal_xy"] == st["goal_xy"]).all()
    assert (_np(env.obs) == obs_ref).all()
    assert (f32_bits(h["reward"]) == f32_bits(np.float32(-0.0))).all()
    # masked second reset after some steps: only masked envs change
    env.step(torch.randint(0, 4, (N,), dtype=torch.int32))
    mask = np.random.RandomState(3).rand(N) < 0.3
    h1 = {k: np.array(v, copy=True) for k, v in env.host_state().items()}
    env.reset(mask=torch.from_numpy(mask))
    O.reset(p, np.ascontiguousarray(lay), mask.astype(np.uint8), seed, 1, h1["ball_xy"],
            h1["goal_xy"] if variant == "v3" else None, h1["step_count"], h1["reward"], h1["done"], obs_ref,
            env_base=77)
    h2 = env.host_state()
    for k in ("ball_xy", "goal_xy", "step_count", "done"):
        assert (h2[k] == h1[k]).all(), k
    assert (f32_bits(h2["reward"]) == f32_bits(h1["reward"])).all()
    assert (h2["step_count"][mask] == 0).all() and (h2["step_count"][~mask] == 1).all()
    assert (_np(env.obs) == obs_ref).all()


def test_reset_placement_is_uniform_over_accepted_cells():
    lay = L.to_codes(L.V0_GRID_12)
    N = 1 << 18
    env = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=9)
    b = _np(env.ball_xy)
    cells = b[:, 0] * 12 + b[:, 1]
    ok = np.flatnonzero((lay.reshape(-1) != ord("W")) & (lay.reshape(-1) != ord("X")))
    counts = np.bincount(cells, minlength=144)
    assert counts[np.setdiff1d(np.arange(144), ok)].sum() == 0       # never a wall, never 'X'
    expect = N / len(ok)
    chi2 = ((counts[ok] - expect) ** 2 / expect).sum()
    assert chi2 < 2.0 * len(ok)                                       # loose: ~len(ok) expected


# ----------------------------------------------------------------------------------------
# 4b. fused auto-reset == reset(mask=done) then step, bit for bit (incl. the Philox epochs)
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["v0", "v3"])
@pytest.mark.parametrize("shared,G,hint", [(True, 8, 0), (True, 11, 0), (True, 12, 0), (True, 9, 0), (False, 8, 0), (False, 11, 0),
                                           (False, 32, 0), (False, 16, 0),
                                           # every compile-time size of the shared kernel (spawn cells as 1 / 2 / 4 / 6 / 16 ballots),
                                           # the 8x8 workgroup kernel (bit 8), 11x11 at 32 and 64 envs per workgroup, several chunks
                                           (True, 14, 0), (True, 18, 0), (True, 32, 0), (True, 8, 0x100), (True, 8, 0x123),
                                           (True, 11, 0x818), (True, 11, 0x423), (True, 11, 0x832), (True, 12, 0x832),
                                           (True, 14, 0x815), (True, 18, 0x824), (True, 14, 0x422), (True, 12, 0xc18), (True, 11, 0xc25),
                                           (True, 32, 0x815), (True, 32, 0x422), (True, 8, 0x914), (True, 8, 0x524),
                                           (True, 9, 0xc15), (True, 20, 0x415), (True, 20, 0x823), (True, 5, 0xc00), (True, 12, 0xc25), (True, 18, 0x816)])   # other G at 16 / 256 / 64 envs   # 32x32 at 4 / 8, 8x8 workgroup kernel at 64 / 128 envs   # 14x14 / 18x18 at 16 envs per workgroup
def test_fused_autoreset_equals_reset_then_step(variant, shared, G, hint):
    N, T, seed = 2500, 60, 21
    kw = dict(variant=variant, seed=seed, step_limit=7, env_base=1000)   # short episodes: many resets
    if shared:
        lay = bordered_random_layouts(1, G, 300 + G)[0]
        fused, split = PKG.LmazeVecEnv(N, layout=lay, **kw), PKG.LmazeVecEnv(N, layout=lay, **kw)
    else:
        lay = bordered_random_layouts(N, G, 300 + G)
        fused, split = PKG.LmazeVecEnv(N, per_env_layouts=lay, **kw), PKG.LmazeVecEnv(N, per_env_layouts=lay, **kw)
    fused.params.launch_hint = hint
    rs = np.random.RandomState(G)
    n_resets = 0
    for t in range(T):
        a = torch.from_numpy(rs.randint(0, 4, N).astype(np.int32))
        n_resets += int(split.done.sum().item())
        split.reset(mask=split.done)
        split.step(a)
        fused.step(a, auto_reset=True)
        hf, hs = fused.host_state(), split.host_state()
        for k in hf:
            assert (hf[k].view(np.uint8) == hs[k].view(np.uint8)).all(), (k, t)
        assert (fused.obs == split.obs).all(), t
    assert n_resets > N          # every env went through several episodes
    # rollout() is the same loop
    acts = torch.randint(0, 4, (5, N), dtype=torch.int32, device="cuda")
    fused.rollout(acts, auto_reset=True)
    for t in range(5):
        split.reset(mask=split.done)
        split.step(acts[t])
    assert (fused.obs == split.obs).all() and (fused.ball_xy == split.ball_xy).all()


# ----------------------------------------------------------------------------------------
# 5. reference-layout render vs oracle, incl. shapes whose C*S*S is not a multiple of 4
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("G,E,cmask", [(12, 7, (1, 2, 4, 8)), (11, 7, (1, 2, 4, 8)), (18, 4, (8, 1, 4)),
                                       (5, 7, (1, 2, 4, 8, 3)), (5, 7, (8, 4, 1)), (9, 3, (1, 2, 4)),
                                       (32, 7, (1, 2, 4, 8)), (8, 1, (1,)), (8, 7, (1, 2, 4, 8)), (7, 7, (4, 1, 256)),
                                       (63, 5, (1, 2, 4)), (64, 16, (2,)), (3, 16, (1, 8)),
                                       (11, 1, (1, 2, 4, 8)), (32, 1, (1, 2, 4, 8)), (64, 1, (2,)), (3, 1, (1, 8)),
                                       (7, 1, (4, 1, 256)), (5, 1, (1, 2, 4, 8, 3, 5, 6, 15)), (12, 1, (8, 1, 4))])
def test_render_expanded_matches_oracle(G, E, cmask):
    """Specialised shapes, the generic kernel, ragged C*S*S, and the shapes too large for the LDS table."""
    abi = importlib.import_module("gym-lmaze_amd._abi")
    import ctypes as C
    N = 37 if G * E < 300 else 3
    obs = np.random.RandomState(G * E).randint(0, 16 if max(cmask) < 16 else 1024, (N, G, G)).astype(np.int32)
    ref = O.render_expanded(obs, G, E, cmask)
    d_obs = torch.from_numpy(obs).cuda()
    out = torch.full((N, len(cmask), G * E, G * E), -1.0, dtype=torch.float32, device="cuda")
    m = (C.c_int32 * len(cmask))(*cmask)
    rc = abi.lib.lmaze_render_expanded(d_obs.data_ptr(), G, E, m, len(cmask), out.data_ptr(), N,
                                       torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert (_np(out).view(np.uint32) == ref.view(np.uint32)).all()


def test_empty_batch_and_bad_arguments_on_device():
    """n = 0 is a no-op on every entry point; rejected arguments launch nothing and report why."""
    import ctypes as C
    abi = importlib.import_module("gym-lmaze_amd._abi")
    env = PKG.LmazeVecEnv(8, variant="v0")
    st = torch.cuda.current_stream().cuda_stream
    before = env._state.clone()
    a = torch.zeros(8, dtype=torch.int32, device="cuda")
    assert abi.lib.lmaze_step_v0(env._pp, env._p_layout, a.data_ptr(), env._p_ball, env._p_step, env._p_reward,
                                 env._p_done, env._p_gc, env._p_obs, 0, st) == 0
    assert abi.lib.lmaze_reset(env._pp, env._p_layout, None, 1, 1, 0, env._p_ball, None, env._p_step, env._p_reward,
                               env._p_done, env._p_obs, 0, st) == 0
    m = (C.c_int32 * 4)(1, 2, 4, 8)
    assert abi.lib.lmaze_render_expanded(env._p_obs, 12, 7, m, 4, env._p_obs, 0, st) == 0
    torch.cuda.synchronize()
    assert (env._state == before).all()
    # misaligned obs pointer -> LMAZE_E_ALIGN, unknown layout mode -> LMAZE_E_LAYOUT; nothing runs
    assert abi.lib.lmaze_step_v0(env._pp, env._p_layout, a.data_ptr(), env._p_ball, env._p_step, env._p_reward,
                                 env._p_done, env._p_gc, env._p_obs + 4, 8, st) == -6
    bad = abi.make_params(abi.VARIANT_V0, 12, 7, 100, -1.0, -0.01, 100.0)
    assert abi.lib.lmaze_step_v0(C.byref(bad), env._p_layout, a.data_ptr(), env._p_ball, env._p_step, env._p_reward,
                                 env._p_done, env._p_gc, env._p_obs, 8, st) == -4
    torch.cuda.synchronize()
    assert (env._state == before).all()
    f = PKG.LmazeFovealVecEnv(4, variant="v2")
    assert abi.lib.lmaze_foveal_step(f._pp, f._p_layouts, a.data_ptr(), f._pb, 0, st) == 0
    assert abi.lib.lmaze_foveal_step(f._pp, f._p_layouts, None, f._pb, 4, st) == -1


# ----------------------------------------------------------------------------------------
# 9. the placement switches of the drop-in classes (RANDOM_BALL / RANDOM_GOAL flipped after
#    construction, as a user of the reference does) against reference recordings.  The same fixtures
#    also go through the plain replays above (golden_files("v0_") / ("v3_") / ("v2_")).
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,vid", [("v0_fixed_start_seed0", "v0"), ("v3_fixed_start_goal_seed1", "v3"),
                                      ("v3_fixed_goal_seed2", "v3"), ("v2_fixed_goal_seed3", "v2"),
                                      ("v2_fixed_start_seed4", "v2")])
def test_dropin_placement_switches(name, vid):
    g = load_golden(name)
    import gym_lmaze
    random.seed(int(g["seed"]))
    np.random.seed(int(g["seed"]))
    env = gym_lmaze.make("lmaze-" + vid)             # the constructor's own reset() draws too
    env.RANDOM_BALL = bool(g["random_ball"])
    if vid != "v0":
        env.RANDOM_GOAL = bool(g["random_goal"])
    if vid != "v2":
        random.seed(int(g["seed"]))                  # the v0 / v3 recordings re-seed after construction
    need_reset, n_reset = True, 0
    for t in range(len(g["actions"])):
        if need_reset:
            o = env.reset()
            assert obs_hash(np.ascontiguousarray(o)) == g["reset_hash"][n_reset], (name, t)
            assert (env.ball_x0, env.ball_y0) == tuple(g["ball_before"][t]), (name, t)
            if vid != "v0":
                assert (env.goal_x, env.goal_y) == tuple(g["goal_before"][t]), (name, t)
            n_reset += 1
            need_reset = False
        a = int(g["actions"][t])
        arg = (str(a) if 0 <= a <= 3 else a) if vid == "v3" else a
        o, r, d, _ = env.step(arg)
        assert r == g["reward"][t] and d == bool(g["done"][t]), (name, t)
        assert obs_hash(np.ascontiguousarray(o)) == g["obs_hash"][t], (name, t)
        assert (env.ball_x0, env.ball_y0) == tuple(g["ball"][t])
        need_reset = d
    assert n_reset == len(g["reset_hash"]) and n_reset >= 3


def test_bandwidth_probe_fills_and_copies():
    abi = importlib.import_module("gym-lmaze_amd._abi")
    st = torch.cuda.current_stream().cuda_stream
    n = (1 << 20) + 4                                   # 16-byte units, ragged against the 256-thread block
    dst = torch.full((n * 4,), -1, dtype=torch.int32, device="cuda")
    assert abi.lib.lmaze_bandwidth_probe(None, dst.data_ptr(), n * 16, st) == 0
    got = dst.view(n, 4).cpu().numpy()
    assert (got[:, 0] == np.arange(n)).all() and (got[:, 1:] == [1, 2, 3]).all()
    src = torch.randint(-2 ** 31, 2 ** 31 - 1, (n * 4,), dtype=torch.int32, device="cuda")
    assert abi.lib.lmaze_bandwidth_probe(src.data_ptr(), dst.data_ptr(), n * 16, st) == 0
    assert torch.equal(src, dst)
    assert abi.lib.lmaze_bandwidth_probe(None, dst.data_ptr(), 24, st) == -5          # not a multiple of 16
    assert abi.lib.lmaze_bandwidth_probe(None, dst.data_ptr() + 4, 16, st) == -6      # misaligned
    assert abi.lib.lmaze_bandwidth_probe(None, None, 16, st) == -1


def test_render_expanded_output_not_on_a_cache_line():
    """The ABI asks for 16-byte alignment only; the aligned-stretch kernels need 64 and must step aside."""
    abi = importlib.import_module("gym-lmaze_amd._abi")
    import ctypes as C
    N, G, E, cmask = 21, 11, 7, (1, 2, 4, 8)
    obs = np.random.RandomState(3).randint(0, 16, (N, G, G)).astype(np.int32)
    ref = O.render_expanded(obs, G, E, cmask)
    d_obs = torch.from_numpy(obs).cuda()
    buf = torch.full((N * 4 * 77 * 77 + 4,), -1.0, dtype=torch.float32, device="cuda")
    out = buf[4:]                                                   # 16 bytes past a cache line
    m = (C.c_int32 * 4)(*cmask)
    st = torch.cuda.current_stream().cuda_stream
    assert abi.lib.lmaze_render_expanded(d_obs.data_ptr(), G, E, m, 4, out.data_ptr(), N, st) == 0
    assert (_np(out).view(np.uint32) == ref.reshape(-1).view(np.uint32)).all() and float(buf[3]) == -1.0
    planes = np.random.RandomState(4).rand(N, 5, 5, 5).astype(np.float32)
    want = O.expand_planes(planes, 7)
    buf2 = torch.full((N * 5 * 35 * 35 + 4,), -1.0, dtype=torch.float32, device="cuda")
    d_planes = torch.from_numpy(planes).cuda()
    assert abi.lib.lmaze_expand_planes(d_planes.data_ptr(), 5, 5, 7, buf2[4:].data_ptr(), N, st) == 0
    assert (_np(buf2[4:]).view(np.uint32) == want.reshape(-1).view(np.uint32)).all() and float(buf2[3]) == -1.0


def test_package_imported_before_torch_still_finds_the_device():
    """A user of the reference writes `import gym_lmaze` first.  Both torch and liblmaze_hip.so bring a HIP
    runtime; the binding must make them share one (a fresh interpreter: this process already has torch)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import contextlib, io\n"
            "import gym_lmaze\n"
            "assert 'torch' in sys.modules\n"
            "with contextlib.redirect_stdout(io.StringIO()):\n"
            "    env = gym_lmaze.make('lmaze-v0')\n"
            "o, r, d, a = env.step(1)\n"
            "print('OK', o.shape, r, d, a)\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK (4, 84, 84)" in out.stdout, out.stderr[-2000:]


def test_random_shapes_fuzz():
    """Forty seeded random (variant, G, N, layout mode) combinations through the same step-by-step
    comparison, to reach grid sizes and batch sizes the parametrised cases above do not name."""
    rs = np.random.RandomState(20261004)
    for case in range(40):
        variant = "v3" if rs.rand() < 0.5 else "v0"
        shared = rs.rand() < 0.5
        G = int(rs.randint(4, 65)) if shared else int(rs.randint(4, 41))
        N = int(rs.choice([1, 2, 7, 64, 65, 130, 257, 1000, 2049]))
        if not shared:
            N = min(N, 600)
        T = 12 if G <= 24 else 5
        try:
            _compare_rollout(variant, N, G, T, shared=shared, seed=1000 + case, action_hi=9)
        except AssertionError as e:
            raise AssertionError("case %d: %s G=%d N=%d shared=%s: %s" % (case, variant, G, N, shared, e))


def test_fused_autoreset_fuzz():
    """Ten more seeded (variant, layout mode, G) combinations of the fused auto-reset comparison."""
    rs = np.random.RandomState(77)
    for case in range(10):
        variant = "v3" if rs.rand() < 0.5 else "v0"
        shared = bool(rs.rand() < 0.5)
        G = int(rs.randint(5, 41))
        test_fused_autoreset_equals_reset_then_step(variant, shared, G, 0)


# ----------------------------------------------------------------------------------------
# 8. the wave-autonomous 8x8 kernel (small shared-layout batches, BASELINE config 2) against the workgroup/LDS
#    kernel it replaces there (LmazeParams.launch_hint bit 8 keeps the latter) and against the oracle
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["v0", "v3"])
@pytest.mark.parametrize("N,hint", [(1, 0), (63, 0x10), (64, 0x10), (65, 0x20), (130, 0x30), (3003, 0), (3003, 0x10), (3003, 0x30),
                                    (1, 0x24), (65, 0x12), (130, 0x34), (3003, 0x22), (3003, 0x14)])   # + 2 / 4 waves per workgroup
def test_wave_autonomous_8x8_kernel_equals_the_lds_kernel_and_the_oracle(variant, N, hint):
    lay = bordered_random_layouts(1, 8, 41 + N)[0]
    kw = dict(variant=variant, layout=lay, seed=5, step_limit=9, env_base=3)
    wave, lds = PKG.LmazeVecEnv(N, **kw), PKG.LmazeVecEnv(N, **kw)
    wave.params.launch_hint = hint           # envs per wave: 64 / 32 / 16 (0 = the default)
    lds.params.launch_hint = 0x100
    assert (wave.obs == lds.obs).all() and (wave._state == lds._state).all()          # constructor: reset + planes
    p, st = _oracle_state(wave, variant, lay)
    obs_ref = np.zeros((N, 8, 8), np.int32)
    lay_c = np.ascontiguousarray(lay)
    rs = np.random.RandomState(N)
    epoch = wave._epoch
    for t in range(50):
        a = np.where(rs.rand(N) < 0.85, rs.randint(0, 4, N), rs.randint(-1, 6, N)).astype(np.int32)
        ta = torch.from_numpy(a)
        auto = t % 3 != 2                     # fused auto-reset on most steps, plain ones in between
        render = t % 7 != 6                   # and some transition-only steps (obs = NULL)
        for e in (wave, lds):
            e.step(ta, render=render, auto_reset=auto)
        if auto:
            O.reset(p, lay_c, st["done"].copy(), 5, epoch, st["ball_xy"], st["goal_xy"] if variant == "v3" else None,
                    st["step_count"], st["reward"], st["done"], None, env_base=3)
            epoch += 1
        if variant == "v3":
            O.step_v3(p, lay_c, a, st["ball_xy"], st["goal_xy"], st["step_count"], st["reward"], st["done"], obs_ref)
        else:
            O.step_v0(p, lay_c, a, st["ball_xy"], st["step_count"], st["reward"], st["done"], st["goal_count"], obs_ref)
        hw, hl = wave.host_state(), lds.host_state()
        for k in hw:
            assert (hw[k].view(np.uint8) == hl[k].view(np.uint8)).all(), (k, t)
        for k in ("ball_xy", "step_count", "goal_count", "done") + (("goal_xy",) if variant == "v3" else ()):
            assert (hw[k] == st[k]).all(), (k, t)
        assert (f32_bits(hw["reward"]) == f32_bits(st["reward"])).all(), t
        if render:
            assert (wave.obs == lds.obs).all() and (_np(wave.obs) == obs_ref).all(), t
        else:
            assert (_np(wave.observe()) == obs_ref).all() and (_np(lds.observe()) == obs_ref).all(), t
    assert wave._epoch == lds._epoch == epoch


# ---------------------------------------------------------------- T steps in one launch (lmaze_rollout)
@pytest.mark.parametrize("variant", ["v0", "v3"])
@pytest.mark.parametrize("auto_reset", [False, True])
@pytest.mark.parametrize("T", [1, 7, 256])
def test_rollout_in_one_launch_equals_T_step_calls(variant, auto_reset, T):
    """lmaze_rollout on BASELINE's C2 shape (65 536 x 8x8, the rollout kernel: a wave keeps its 64 envs in registers
    across the T steps) and on ragged batches: state, planes and every step's reward / done row are bit-identical to T
    calls of lmaze_step_* (with the fused reset: the same placements, epoch + t)."""
    lay = PKG.layouts.GRID_8_BORDERED
    for N in ((65536, 1000, 70) if T != 256 else (65536,)):
        one = PKG.LmazeVecEnv(N, variant=variant, layout=lay, seed=11, env_base=3)
        ref = PKG.LmazeVecEnv(N, variant=variant, layout=lay, seed=11, env_base=3)
        what = PKG._abi.describe_step(one.params, N, auto_reset)
        assert "step_shared_wave8_kernel" in what
        gen = torch.Generator(device="cuda").manual_seed(5 + T)
        acts = torch.randint(-1, 6, (T, N), dtype=torch.int32, device="cuda", generator=gen)    # incl. out-of-range ids
        obs, rew, done, rew_t, done_t = one.rollout(acts, auto_reset=auto_reset, trajectory=True)
        for t in range(T):
            o, r, d, _ = ref.step(acts[t], auto_reset=auto_reset)
            assert (rew_t[t].view(torch.int32) == r.view(torch.int32)).all(), (N, t)
            assert (done_t[t] == d).all(), (N, t)
        h1, h2 = one.host_state(), ref.host_state()
        for k in h1:
            assert (np.ascontiguousarray(h1[k]).view(np.uint8) == np.ascontiguousarray(h2[k]).view(np.uint8)).all(), (k, N)
        assert (one.obs == ref.obs).all() and one._epoch == ref._epoch
        if auto_reset and T >= 7:
            assert int(one.goal_count.sum().item()) == int(ref.goal_count.sum().item())


@pytest.mark.parametrize("variant,G,N,T", [("v0", 11, 5000, 9), ("v3", 11, 70000, 40), ("v0", 12, 1000, 130), ("v3", 18, 300, 25),
                                           ("v0", 33, 100, 12), ("v3", 5, 17, 60), ("v0", 11, 65536, 16), ("v3", 64, 33, 7)])
@pytest.mark.parametrize("auto_reset", [False, True])
def test_rollout_of_any_on_die_shared_batch_is_one_launch(variant, G, N, T, auto_reset):
    """Shared layouts of any grid size whose planes stay on-die: lmaze_rollout is ONE launch of rollout_shared_kernel (a
    workgroup keeps its envs' state in registers across the T steps) -- state, planes, goal counts and every step's reward /
    done row bit-identical to T calls of lmaze_step_* (fused reset: the same placements, epoch + t); even and odd G, ragged
    batches, out-of-range action ids."""
    lay = PKG.layouts.open_room(G, (G // 2, G // 2)) if G != 12 else PKG.layouts.V0_GRID_12
    one = PKG.LmazeVecEnv(N, variant=variant, layout=lay, seed=2, env_base=5)
    ref = PKG.LmazeVecEnv(N, variant=variant, layout=lay, seed=2, env_base=5)
    gen = torch.Generator(device="cuda").manual_seed(G * 100 + T)
    acts = torch.randint(-1, 6, (T, N), dtype=torch.int32, device="cuda", generator=gen)
    obs, rew, done, rew_t, done_t = one.rollout(acts, auto_reset=auto_reset, trajectory=True)
    for t in range(T):
        o, r, d, _ = ref.step(acts[t], auto_reset=auto_reset)
        assert (rew_t[t].view(torch.int32) == r.view(torch.int32)).all(), t
        assert (done_t[t] == d).all(), t
    h1, h2 = one.host_state(), ref.host_state()
    for k in h1:
        assert (np.ascontiguousarray(h1[k]).view(np.uint8) == np.ascontiguousarray(h2[k]).view(np.uint8)).all(), k
    assert (one.obs == ref.obs).all() and one._epoch == ref._epoch
    assert (one.goal_count == ref.goal_count).all()
    # a second rollout continues where the first ended (done flags carried across calls)
    one.rollout(acts[: min(T, 5)], auto_reset=auto_reset)
    for t in range(min(T, 5)):
        ref.step(acts[t], auto_reset=auto_reset)
    assert (one.obs == ref.obs).all() and (one.ball_xy == ref.ball_xy).all() and (one.step_count == ref.step_count).all()


@pytest.mark.parametrize("variant,G,N,T", [("v0", 11, 3000, 9), ("v3", 11, 20000, 30), ("v0", 32, 1000, 12), ("v3", 13, 70, 50),
                                           ("v0", 32, 70001, 6), ("v3", 64, 21, 5), ("v0", 5, 130, 40)])
@pytest.mark.parametrize("auto_reset", [False, True])
def test_rollout_of_per_env_batches_is_one_launch(variant, G, N, T, auto_reset):
    """Per-env layouts, on-die and streaming sizes (70 001 x 32x32: 287 MB of planes): lmaze_rollout is ONE launch of
    rollout_perenv_kernel (the workgroup's layouts in LDS for all T steps; a done env is re-placed by a whole wave on its own
    maze, the per-env step kernels' rule) -- bit-identical to T calls of lmaze_step_* incl. every step's reward / done row."""
    lays = PKG.layouts.random_walled(N, G, torch.device("cuda"), seed=7 + G)
    one = PKG.LmazeVecEnv(N, variant=variant, per_env_layouts=lays, seed=2, env_base=11)
    ref = PKG.LmazeVecEnv(N, variant=variant, per_env_layouts=lays, seed=2, env_base=11)
    gen = torch.Generator(device="cuda").manual_seed(G * 100 + T)
    acts = torch.randint(-1, 6, (T, N), dtype=torch.int32, device="cuda", generator=gen)
    obs, rew, done, rew_t, done_t = one.rollout(acts, auto_reset=auto_reset, trajectory=True)
    for t in range(T):
        o, r, d, _ = ref.step(acts[t], auto_reset=auto_reset)
        assert (rew_t[t].view(torch.int32) == r.view(torch.int32)).all() and (done_t[t] == d).all(), t
    h1, h2 = one.host_state(), ref.host_state()
    for k in h1:
        assert (np.ascontiguousarray(h1[k]).view(np.uint8) == np.ascontiguousarray(h2[k]).view(np.uint8)).all(), k
    assert (one.obs == ref.obs).all() and one._epoch == ref._epoch and (one.goal_count == ref.goal_count).all()
    one.rollout(acts[: min(T, 4)], auto_reset=auto_reset)
    for t in range(min(T, 4)):
        ref.step(acts[t], auto_reset=auto_reset)
    assert (one.obs == ref.obs).all() and (one.ball_xy == ref.ball_xy).all() and (one.step_count == ref.step_count).all()


@pytest.mark.parametrize("variant,G,N", [("v0", 11, 1 << 20), ("v3", 18, 300001)])
def test_rollout_of_a_streaming_shared_batch_is_one_launch_too(variant, G, N):
    """Batches far beyond the caches (BASELINE's 1M x 11x11: 507 MB of planes): still one launch of rollout_shared_kernel,
    still bit-identical to T step calls with fused resets, trajectory rows included."""
    lay = PKG.layouts.open_room(G, (G // 2, G // 2))
    one = PKG.LmazeVecEnv(N, variant=variant, layout=lay, seed=4, env_base=1 << 33)
    ref = PKG.LmazeVecEnv(N, variant=variant, layout=lay, seed=4, env_base=1 << 33)
    T = 6
    acts = torch.randint(-1, 5, (T, N), dtype=torch.int32, device="cuda")
    for env in (one, ref):                      # some envs already done when the rollout starts
        env.step_count.fill_(env.params.step_limit - 2)
    obs, rew, done, rew_t, done_t = one.rollout(acts, auto_reset=True, trajectory=True)
    for t in range(T):
        o, r, d, _ = ref.step(acts[t], auto_reset=True)
        assert (rew_t[t].view(torch.int32) == r.view(torch.int32)).all() and (done_t[t] == d).all(), t
    assert (one.obs == ref.obs).all() and (one.ball_xy == ref.ball_xy).all() and (one.step_count == ref.step_count).all()
    assert one._epoch == ref._epoch and int(done.sum().item()) == int(ref.done.sum().item())


# ---------------------------------------------------------------- narrow observation (uint8 planes)
@pytest.mark.parametrize("variant,G,N", [("v0", 11, 5000), ("v0", 11, 64), ("v3", 11, 777), ("v0", 8, 4096), ("v0", 12, 1025),
                                         ("v3", 18, 300), ("v0", 5, 1000), ("v0", 4, 33), ("v0", 33, 130), ("v3", 64, 17)])
def test_u8_observation_is_the_int32_planes_narrowed(variant, G, N):
    """obs_dtype='u8' (lmaze_step_u8 / lmaze_observe_u8): state and planes equal the int32 mode's on every step, with
    the fused reset, after masked resets, for even and odd G and ragged batches (a 16-byte store holds 16 cells and
    straddles envs at every odd G)."""
    lay = PKG.layouts.open_room(G, (G // 2, G // 2))
    wide = PKG.LmazeVecEnv(N, variant=variant, layout=lay, seed=21, env_base=7)
    narrow = PKG.LmazeVecEnv(N, variant=variant, layout=lay, seed=21, env_base=7, obs_dtype="u8")
    assert narrow.obs.dtype == torch.uint8 and tuple(narrow.obs.shape) == (N, G, G)
    assert "step_shared_u8_kernel" not in PKG._abi.describe_step(wide.params, N)
    assert (narrow.obs == wide.obs.to(torch.uint8)).all()                      # reset planes
    gen = torch.Generator(device="cuda").manual_seed(G * 1000 + N)
    for t in range(60):
        a = torch.randint(-1, 6, (N,), dtype=torch.int32, device="cuda", generator=gen)
        ar = t % 3 != 0
        ow, rw, dw, _ = wide.step(a, auto_reset=ar)
        on, rn, dn, _ = narrow.step(a, auto_reset=ar)
        assert (on == ow.to(torch.uint8)).all(), t
        assert (rn.view(torch.int32) == rw.view(torch.int32)).all() and (dn == dw).all(), t
        if t % 17 == 5:                                                         # masked reset: only those planes change
            m = torch.rand(N, device="cuda", generator=gen) < 0.3
            wide.reset(mask=m)
            narrow.reset(mask=m)
            assert (narrow.obs == wide.obs.to(torch.uint8)).all(), t
    hw, hn = wide.host_state(), narrow.host_state()
    for k in hw:
        assert (np.ascontiguousarray(hw[k]).view(np.uint8) == np.ascontiguousarray(hn[k]).view(np.uint8)).all(), k
    assert (narrow.expanded() == wide.expanded()).all()


def test_u8_observation_at_c3_size_and_refusals():
    N, G = 1 << 20, 11
    lay = PKG.layouts.open_room(G, (5, 5))
    wide = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=1)
    narrow = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=1, obs_dtype="u8")
    acts = torch.randint(0, 4, (12, N), dtype=torch.int32, device="cuda")
    for t in range(12):
        wide.step(acts[t], auto_reset=True)
        narrow.step(acts[t], auto_reset=True)
    assert (narrow.obs == wide.obs.to(torch.uint8)).all() and (narrow.ball_xy == wide.ball_xy).all()
    narrow.rollout(acts[:4], auto_reset=True)                                   # T step launches (no one-launch form)
    wide.rollout(acts[:4], auto_reset=True)
    assert (narrow.obs == wide.obs.to(torch.uint8)).all() and narrow._epoch == wide._epoch
    with pytest.raises(ValueError):
        PKG.LmazeVecEnv(8, variant="v0", per_env_layouts=np.stack([PKG.layouts.to_codes(lay)] * 8), obs_dtype="u8")
