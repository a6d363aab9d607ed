"""GPU: BASELINE.json's full sizes.  A few steps against the threaded C oracle, then
size-independent properties over a longer rollout, and shard equivalence (section 8(e))."""
import importlib

import numpy as np
import pytest
import torch

import oracle_lib as O
from helpers import f32_bits

pytestmark = pytest.mark.gpu

PKG = importlib.import_module("gym-lmaze_amd")
L = PKG.layouts


def _np(t):
    return t.detach().cpu().numpy()


def _check_invariants(env, layout_bits):
    """Every env: exactly one ball bit, at ball_xy; static bits equal the layout's."""
    obs = env.obs
    N, G = env.num_envs, env.grid
    ball = (obs & 1)
    assert bool((ball.flatten(1).sum(dim=1) == 1).all())
    idx = (env.ball_xy[:, 0].long() * G + env.ball_xy[:, 1].long())
    assert bool((ball.flatten(1).gather(1, idx[:, None]) == 1).all())
    assert bool(((obs & ~1) == layout_bits).all())


def test_c3_1m_11x11_vs_oracle_and_properties():
    N, G = 1 << 20, 11
    lay = L.to_codes(L.open_room(G, (5, 5)))
    env = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=1)
    st = {k: np.array(v, copy=True) for k, v in env.host_state().items()}
    p = O.params(O.VARIANT_V0, G, O.LAYOUT_SHARED)
    ref = np.zeros((N, G, G), np.int32)
    gen = torch.Generator(device="cuda").manual_seed(1)
    for t in range(3):
        a = torch.randint(0, 4, (N,), dtype=torch.int32, device="cuda", generator=gen)
        obs, _, _, _ = env.step(a)
        O.step_v0(p, lay, _np(a), st["ball_xy"], st["step_count"], st["reward"], st["done"], st["goal_count"], ref)
        assert (_np(obs) == ref).all()
        h = env.host_state()
        assert (h["ball_xy"] == st["ball_xy"]).all() and (h["done"] == st["done"]).all()
        assert (f32_bits(h["reward"]) == f32_bits(st["reward"])).all()
    bits = torch.from_numpy(np.where(lay == ord("W"), 2, np.where(lay == ord("X"), 4, np.where(lay == ord("B"), 8, 0)))
                            .astype(np.int32)).cuda()
    for t in range(120):
        a = torch.randint(0, 4, (N,), dtype=torch.int32, device="cuda", generator=gen)
        _, rew, done, _ = env.step(a)
        if t % 40 == 0 or t == 119:
            _check_invariants(env, bits)
    assert bool((env.step_count == 123).all())
    # done is exactly (reward == 100 or stepCount == 100): no env is at step 100 any more
    assert bool((done == (rew == 100.0)).all())
    # goal hits happened and were counted
    assert int(env.goal_count.sum().item()) > 0


def test_c5_1m_32x32_per_env_layouts():
    N, G = 1 << 20, 32
    gen = torch.Generator(device="cuda").manual_seed(7)
    # SURVEY 8(d) C5 as bench.py builds it: 'X' on a uniformly chosen free cell of each maze, the ball on another
    lay = PKG.layouts.random_walled(N, G, torch.device("cuda"), p_wall=0.25, seed=7)
    assert int((lay == ord("X")).sum().item()) == N and bool((lay[:, 0, :] == ord("W")).all())
    xs = torch.nonzero(lay == ord("X"))
    assert len(torch.unique(xs[:, 1] * G + xs[:, 2])) > 800          # spread over the interior, not one pinned cell
    env = PKG.LmazeVecEnv(N, variant="v0", per_env_layouts=lay, seed=2)
    assert not bool((lay.view(N, -1).gather(1, (env.ball_xy[:, 0] * G + env.ball_xy[:, 1]).long()[:, None]) != ord("B")).any())
    lay_np = _np(lay)
    st = {k: np.array(v, copy=True) for k, v in env.host_state().items()}
    # reset parity at full size (same Philox draws; per-env accepted-cell scan)
    z = {k: np.zeros_like(v) for k, v in st.items()}
    z["goal_xy"][...] = st["goal_xy"]
    p = O.params(O.VARIANT_V0, G, O.LAYOUT_PER_ENV)
    O.reset(p, lay_np, None, 2, 0, z["ball_xy"], None, z["step_count"], z["reward"], z["done"], None)
    assert (z["ball_xy"] == st["ball_xy"]).all()
    ref = np.zeros((N, G, G), np.int32)
    for t in range(2):
        a = torch.randint(0, 4, (N,), dtype=torch.int32, device="cuda", generator=gen)
        obs, _, _, _ = env.step(a)
        O.step_v0(p, lay_np, _np(a), st["ball_xy"], st["step_count"], st["reward"], st["done"], st["goal_count"], ref)
        assert (_np(obs) == ref).all()
        assert (env.host_state()["ball_xy"] == st["ball_xy"]).all()
    bits = torch.where(lay == ord("W"), 2, torch.where(lay == ord("X"), 4, torch.where(lay == ord("B"), 8, 0))).to(torch.int32)
    for t in range(30):
        env.step(torch.randint(0, 4, (N,), dtype=torch.int32, device="cuda", generator=gen))
    _check_invariants(env, bits)


@pytest.mark.parametrize("variant,G,N", [("v0", 11, 460_000), ("v3", 12, 400_003), ("v0", 18, 170_001)])
def test_per_env_lds_kernel_streaming_regime_against_the_oracle(variant, G, N):
    """The LDS-tiled per-env kernel (every G whose G*G is not a multiple of 256) once its observation exceeds 192 MiB:
    non-temporal stores, 8-KiB layout tiles, ragged last workgroup -- fused reset included, every step against the oracle."""
    from helpers import bordered_random_layouts
    lay_np = bordered_random_layouts(N, G, 900 + G)
    env = PKG.LmazeVecEnv(N, variant=variant, per_env_layouts=torch.from_numpy(lay_np), seed=4, step_limit=5, env_base=77)
    assert env.obs.numel() * 4 > (192 << 20)
    st = {k: np.array(v, copy=True) for k, v in env.host_state().items()}
    p = O.params(O.VARIANT_V3 if variant == "v3" else O.VARIANT_V0, G, O.LAYOUT_PER_ENV, env.step_limit, *env.rewards)
    ref = np.zeros((N, G, G), np.int32)
    rs = np.random.RandomState(G)
    epoch = env._epoch
    for t in range(8):
        a = rs.randint(0, 5, N).astype(np.int32)
        mask = np.ascontiguousarray(st["done"])
        O.reset(p, lay_np, mask, 4, epoch, st["ball_xy"], st["goal_xy"] if variant == "v3" else None, st["step_count"],
                st["reward"], st["done"], None, env_base=77)
        epoch += 1
        if variant == "v3":
            O.step_v3(p, lay_np, a, st["ball_xy"], st["goal_xy"], st["step_count"], st["reward"], st["done"], ref)
        else:
            O.step_v0(p, lay_np, a, st["ball_xy"], st["step_count"], st["reward"], st["done"], st["goal_count"], ref)
        obs, _, _, _ = env.step(torch.from_numpy(a), auto_reset=True)
        h = env.host_state()
        for k in ("ball_xy", "step_count", "done"):
            assert (h[k] == st[k]).all(), (k, t)
        assert (f32_bits(h["reward"]) == f32_bits(st["reward"])).all(), t
        assert (_np(obs) == ref).all(), t
    assert env._epoch == epoch


@pytest.mark.parametrize("variant", ["v0", "v3"])
def test_shard_equivalence(variant):
    """8 shards of N/8 (run one after another on this GPU) == one run of N: same reset draws
    (env_base), same transitions, same planes."""
    N, G, T, world = 8 * 4099, 12, 25, 8       # 4099: shards are not workgroup multiples
    lay = L.to_codes(L.V0_GRID_12)
    rs = np.random.RandomState(5)
    acts = rs.randint(-1, 5, (T, N)).astype(np.int32)
    full = PKG.LmazeVecEnv(N, variant=variant, layout=lay, seed=11)
    for t in range(T):
        full.step(torch.from_numpy(acts[t]))
        if t == 12:
            full.reset(mask=full.done)
    hf = full.host_state()
    of = _np(full.obs)
    for r in range(world):
        start, count = PKG.shard_range(N, r, world)
        sh = PKG.LmazeVecEnv(count, variant=variant, layout=lay, seed=11, env_base=start)
        for t in range(T):
            sh.step(torch.from_numpy(acts[t, start:start + count].copy()))
            if t == 12:
                sh.reset(mask=sh.done)
        hs = sh.host_state()
        for k in hs:
            a, b = hs[k], hf[k][start:start + count]
            assert (a.view(np.uint8) == np.ascontiguousarray(b).view(np.uint8)).all(), (k, r)
        assert (_np(sh.obs) == of[start:start + count]).all(), r


def test_graph_capture_replays_steps():
    """The launch path allocates nothing and never syncs, so K steps capture into one hipGraph."""
    N, G, K = 4096, 8, 6
    lay = L.to_codes(L.GRID_8_BORDERED)
    env_a = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=4)
    env_b = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=4)
    acts = torch.randint(0, 4, (K, N), dtype=torch.int32, device="cuda")
    for t in range(K):
        env_a.step(acts[t])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for t in range(K):
                env_b.step_raw(acts[t].data_ptr())
    # capture itself runs nothing: env_b is still at step 0
    assert int(env_b.step_count.max().item()) == 0
    graph.replay()
    torch.cuda.synchronize()
    assert (env_a.obs == env_b.obs).all() and (env_a.ball_xy == env_b.ball_xy).all()
    assert (env_a.step_count == env_b.step_count).all()
    # the host helper does the same, and a replay advances the state again
    env_c = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=4)
    g2 = env_c.capture_rollout(acts)
    g2.replay()
    torch.cuda.synchronize()
    assert (env_c.obs == env_a.obs).all()
    g2.replay()
    for t in range(K):
        env_a.step(acts[t])
    torch.cuda.synchronize()
    assert (env_c.obs == env_a.obs).all() and (env_c.step_count == 2 * K).all()


def test_autotune_keeps_results_and_state():
    """autotune() times real steps under each launch policy, restores the state, and any hint
    gives bit-identical results (the hint only changes occupancy)."""
    N, G = 1 << 19, 11                     # 254 MB of obs: the non-temporal regime (> 192 MiB)
    lay = L.to_codes(L.open_room(G, (5, 5)))
    env = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=6)
    ref = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=6)
    before = env._state.clone()
    t = env.autotune()
    assert set(t) == set(env.CANDIDATES) and env.params.launch_hint in [env.launch_hint_of(*c) for c in t]
    best = min(t, key=t.get)        # the fastest pair, or the default (3, 2) when nothing beats it by more than 1.5 %
    assert env.tuned_policy in (best, env.DEFAULT_POLICY) and (env.tuned_policy == best or t[best] > 0.985 * t[env.DEFAULT_POLICY])
    assert (env._state == before).all()
    acts = torch.randint(0, 4, (4, N), dtype=torch.int32, device="cuda")
    t2 = env.autotune(actions=acts, candidates=((3, 1), (8, 2)), steps=4, warm=8)
    assert set(t2) == {(3, 1), (8, 2)} and (env._state == before).all()
    scratch = torch.empty(32 << 20, dtype=torch.int32, device="cuda")
    t3 = env.autotune(actions=acts, candidates=((3, 2), (8, 1)), steps=4, warm=8, between=lambda: scratch.add_(1))
    assert set(t3) == {(3, 2), (8, 1)} and (env._state == before).all() and all(v > 0 for v in t3.values())
    # workgroups per CU in bits 0-3, chunks per workgroup in bits 4-7: every combination, same results
    for hint in (0, 2, 3, 5, 8, 0x13, 0x23, 0x48, 0xF7, 0x30):
        env.params.launch_hint = hint
        env._state.copy_(before)
        ref._state.copy_(before)
        for k in range(4):
            env.step(acts[k])
            ref.step(acts[k])
        assert (env.obs == ref.obs).all() and (env._state == ref._state).all(), hint


def test_autotune_placement_trials_rehome_the_observation_buffer_only():
    """autotune(placement_trials=K) may move env.obs to another allocation (the fastest of K); state, epoch and the
    planes are what they were, and the rollout afterwards equals that of an env that never tuned."""
    N, G = 1 << 19, 11
    lay = L.to_codes(L.open_room(G, (5, 5)))
    env = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=6)
    ref = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=6)
    acts = torch.randint(0, 4, (6, N), dtype=torch.int32, device="cuda")
    env.step(acts[0]); ref.step(acts[0])
    t = env.autotune(actions=acts, steps=4, warm=8, rounds=1, placement_trials=3)
    assert len(env.placement["trials_ms"]) == 3 and 0 <= env.placement["kept"] < 3 and set(t) == set(env.CANDIDATES)
    assert env._p_obs == env.obs.data_ptr() and (env.obs == ref.obs).all() and (env._state == ref._state).all()
    for k in range(1, 6):
        o1, _, _, _ = env.step(acts[k], auto_reset=True)
        o2, _, _, _ = ref.step(acts[k], auto_reset=True)
        assert o1.data_ptr() == env.obs.data_ptr() and (o1 == o2).all() and (env._state == ref._state).all(), k
    assert (env.expanded()[:64] == ref.expanded()[:64]).all()


def test_episode_stats_match_numpy():
    N, G = 100003, 11
    lay = L.to_codes(L.open_room(G, (5, 5)))
    env = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=8, step_limit=17)
    gen = torch.Generator(device="cuda").manual_seed(3)
    for t in range(40):
        env.step(torch.randint(0, 4, (N,), dtype=torch.int32, device="cuda", generator=gen), auto_reset=True)
        if t in (0, 16, 39):
            s = env.episode_stats()
            h = env.host_state()
            assert s["done"] == int(h["done"].sum())
            assert s["goal_rewards"] == int((h["reward"] == np.float32(100.0)).sum())
            assert s["done_steps"] == int(h["step_count"][h["done"] != 0].sum())
            assert s["goal_count"] == int(h["goal_count"].sum())
    assert s["goal_count"] > 0 and env.episode_stats(all_ranks=True) == s     # no process group: the same numbers
    v3 = PKG.LmazeVecEnv(5000, variant="v3", seed=1)
    v3.step(torch.randint(0, 4, (5000,), dtype=torch.int32, device="cuda"))
    assert v3.episode_stats()["goal_count"] == 0 and v3.episode_stats()["done"] == int(v3.done.sum().item())


@pytest.mark.parametrize("variant,shared,G", [("v0", True, 8), ("v3", True, 11), ("v0", False, 12), ("v3", False, 16),
                                              ("v0", False, 32)])
def test_captured_autoreset_rollout_draws_fresh_placements(variant, shared, G):
    """A captured auto-reset rollout keeps its reset epoch in a device word the launches hand on to each
    other: replays (odd T, so the two words swap roles between replays) and eager steps in between are
    bit-identical to the same steps launched eagerly, i.e. no replay re-uses an earlier replay's draws."""
    from helpers import bordered_random_layouts
    N, T = 3000, 7
    kw = dict(variant=variant, seed=13, step_limit=4, env_base=77)      # short episodes: resets every replay
    if shared:
        lay = bordered_random_layouts(1, G, 40 + G)[0]
        eager, graphed = PKG.LmazeVecEnv(N, layout=lay, **kw), PKG.LmazeVecEnv(N, layout=lay, **kw)
    else:
        lay = bordered_random_layouts(N, G, 40 + G)
        eager, graphed = PKG.LmazeVecEnv(N, per_env_layouts=lay, **kw), PKG.LmazeVecEnv(N, per_env_layouts=lay, **kw)
    acts = torch.randint(0, 4, (T, N), dtype=torch.int32, device="cuda")
    extra = torch.randint(0, 4, (N,), dtype=torch.int32, device="cuda")
    g = graphed.capture_rollout(acts, auto_reset=True)
    assert int(graphed.step_count.max().item()) == 0                   # capture runs nothing

    def same(tag):
        torch.cuda.synchronize()
        he, hg = eager.host_state(), graphed.host_state()
        for k in he:
            assert (he[k].view(np.uint8) == hg[k].view(np.uint8)).all(), (k, tag)
        assert (eager.obs == graphed.obs).all(), tag

    starts = []
    for rep in range(3):
        g.replay()
        eager.rollout(acts, auto_reset=True)
        same(("replay", rep))
        starts.append(_np(graphed.ball_xy).copy())
        if rep == 1:                                                   # an eager launch between two replays
            graphed.step(extra, auto_reset=True)
            eager.step(extra, auto_reset=True)
            same("eager step in between")
    assert not (starts[0] == starts[1]).all() and not (starts[1] == starts[2]).all()
    # the ABI refuses aliased or half-given epoch words
    abi = importlib.import_module("gym-lmaze_amd._abi")
    w = torch.zeros(2, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    args = (eager._pp, eager._p_layout, extra.data_ptr(), eager._p_ball)
    tail = (eager._p_step, eager._p_reward, eager._p_done)
    if variant == "v3":
        call = lambda i, o: abi.lib.lmaze_step_v3_autoreset(*args, eager._p_goal, *tail, eager._p_obs, N, 1, 0, 0, i, o, st)
    else:
        call = lambda i, o: abi.lib.lmaze_step_v0_autoreset(*args, *tail, eager._p_gc, eager._p_obs, N, 1, 0, 0, i, o, st)
    assert call(w.data_ptr(), w.data_ptr()) == -6 and call(None, w.data_ptr()) == -6 and call(w.data_ptr() + 4, None) == -6


@pytest.mark.parametrize("variant", ["v2", "v4"])
def test_foveal_captured_autoreset_steps(variant):
    N, T = 2000, 29                                                    # 3 replays > one 51-step episode
    eager = PKG.LmazeFovealVecEnv(N, variant=variant, seed=3)
    graphed = PKG.LmazeFovealVecEnv(N, variant=variant, seed=3)
    acts = torch.randint(0, 25, (T, N), dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    graphed.begin_replay(0)                                            # allocate the epoch words before capture
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for t in range(T):
                graphed.step(acts[t], auto_reset=True, epoch_slot=t)
    torch.cuda.current_stream().wait_stream(side)
    for rep in range(3):
        graphed.begin_replay(T)
        graph.replay()
        for t in range(T):
            eager.step(acts[t], auto_reset=True)
        torch.cuda.synchronize()
        he, hg = eager.host_state(), graphed.host_state()
        for k in he:
            assert (he[k].view(np.uint8) == hg[k].view(np.uint8)).all(), (k, rep)
        assert (eager.obs.view(torch.int32) == graphed.obs.view(torch.int32)).all(), rep


@pytest.mark.parametrize("G,E,cmask", [(11, 7, (1, 2, 4, 8)), (12, 7, (1, 2, 4, 8)), (18, 4, (8, 1, 4)), (32, 7, (1, 2, 4, 8)),
                                       (11, 1, (1, 2, 4, 8)), (18, 1, (8, 1, 4)), (9, 1, (1,))])
def test_render_expanded_streaming_regime(G, E, cmask):
    """Outputs beyond 192 MiB take the non-temporal, occupancy-capped launch: same floats as the definition
    out[i,c,x*E+xx,y*E+yy] = (obs[i,x,y] & mask[c]) != 0, written here with torch ops."""
    import ctypes as C
    abi = importlib.import_module("gym-lmaze_amd._abi")
    per_env = len(cmask) * (G * E) ** 2 * 4
    N = (200 << 20) // per_env + 3                                     # just past the threshold, ragged count
    obs = torch.randint(0, 16, (N, G, G), dtype=torch.int32, device="cuda")
    out = torch.full((N, len(cmask), G * E, G * E), -1.0, dtype=torch.float32, device="cuda")
    m = (C.c_int32 * len(cmask))(*cmask)
    assert abi.lib.lmaze_render_expanded(obs.data_ptr(), G, E, m, len(cmask), out.data_ptr(), N,
                                         torch.cuda.current_stream().cuda_stream) == 0
    masks = torch.tensor(cmask, dtype=torch.int32, device="cuda")[None, :, None, None]
    step = 256 if E > 1 else 16384
    for lo in range(0, N, step):                                       # in slices: the torch expression is memory-hungry
        want = ((obs[lo:lo + step, None] & masks) != 0).to(torch.float32)
        want = want.repeat_interleave(E, dim=2).repeat_interleave(E, dim=3)
        assert torch.equal(out[lo:lo + step], want), lo


@pytest.mark.parametrize("Cn", [4, 5, 7])
def test_expand_planes_streaming_regime(Cn):
    abi = importlib.import_module("gym-lmaze_amd._abi")
    N = (200 << 20) // (Cn * 35 * 35 * 4) + 5
    planes = torch.rand((N, Cn, 5, 5), dtype=torch.float32, device="cuda")
    out = torch.full((N, Cn, 35, 35), -1.0, dtype=torch.float32, device="cuda")
    assert abi.lib.lmaze_expand_planes(planes.data_ptr(), Cn, 5, 7, out.data_ptr(), N,
                                       torch.cuda.current_stream().cuda_stream) == 0
    want = planes.repeat_interleave(7, dim=2).repeat_interleave(7, dim=3)
    assert torch.equal(out.view(torch.int32), want.view(torch.int32))


@pytest.mark.parametrize("variant,shared,G", [("v0", True, 11), ("v3", True, 12), ("v0", False, 32), ("v3", False, 9)])
def test_long_rollout_with_fused_resets_against_the_oracle(variant, shared, G):
    """3 000 steps (dozens of episodes per env) of the fused auto-reset path against the oracle doing
    reset(mask=done) + step with the same Philox draws; compared every 100 steps and at the end."""
    from helpers import bordered_random_layouts
    N, T, seed, base = 2048, 3000, 5, 123456789
    if shared:
        lay = bordered_random_layouts(1, G, 900 + G)[0]
        env = PKG.LmazeVecEnv(N, variant=variant, layout=lay, seed=seed, env_base=base)
        mode = O.LAYOUT_SHARED
    else:
        lay = bordered_random_layouts(N, G, 900 + G)
        env = PKG.LmazeVecEnv(N, variant=variant, per_env_layouts=lay, seed=seed, env_base=base)
        mode = O.LAYOUT_PER_ENV
    vid = O.VARIANT_V3 if variant == "v3" else O.VARIANT_V0
    p = O.params(vid, G, mode, env.step_limit, *env.rewards)
    st = {k: np.array(v, copy=True) for k, v in env.host_state().items()}
    layc = np.ascontiguousarray(lay)
    ref = np.zeros((N, G, G), np.int32)
    acts = torch.randint(-1, 5, (T, N), dtype=torch.int32, device="cuda")     # some no-op ids among them
    acts_h = acts.cpu().numpy()
    epoch = env._epoch
    goal = st["goal_xy"] if variant == "v3" else None
    episodes = 0
    for t in range(T):
        env.step(acts[t], auto_reset=True)
        mask = np.ascontiguousarray(st["done"])
        episodes += int(mask.sum())
        O.reset(p, layc, mask, seed, epoch + t, st["ball_xy"], goal, st["step_count"], st["reward"], st["done"], None,
                env_base=base)
        if variant == "v3":
            O.step_v3(p, layc, acts_h[t], st["ball_xy"], st["goal_xy"], st["step_count"], st["reward"], st["done"], ref)
        else:
            O.step_v0(p, layc, acts_h[t], st["ball_xy"], st["step_count"], st["reward"], st["done"], st["goal_count"], ref)
        if t % 100 == 99 or t == T - 1:
            h = env.host_state()
            for k in ("ball_xy", "goal_xy", "step_count", "done") + (("goal_count",) if variant == "v0" else ()):
                assert (h[k] == st[k]).all(), (k, t)
            assert (f32_bits(h["reward"]) == f32_bits(st["reward"])).all(), t
            assert (_np(env.obs) == ref).all(), t
    assert episodes > 20 * N


def test_c4_whole_batch_on_one_gpu_8m_envs():
    """BASELINE configs[3] is 8 388 608 x 11x11 over eight GPUs; all of it also fits one MI355X (4.1 GB of
    planes).  Past 2^31 bytes of observation the index arithmetic has to be 64-bit everywhere: invariants on
    every env, and the tail of the batch against the oracle (the last 4 096 envs, same global ids)."""
    N, G, T = 1 << 23, 11, 6
    lay = L.to_codes(L.open_room(G, (5, 5)))
    env = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=9)
    tail = slice(N - 4096, N)
    st = {k: np.array(v[tail], copy=True) for k, v in env.host_state().items()}
    p = O.params(O.VARIANT_V0, G, O.LAYOUT_SHARED)
    ref = np.zeros((4096, G, G), np.int32)
    gen = torch.Generator(device="cuda").manual_seed(3)
    for t in range(T):
        a = torch.randint(0, 4, (N,), dtype=torch.int32, device="cuda", generator=gen)
        env.step(a, auto_reset=(t >= 3))
        a_tail = np.ascontiguousarray(_np(a[tail]))
        if t >= 3:   # nobody is done after 3 steps of a 100-step episode unless the goal was hit
            O.reset(p, lay, np.ascontiguousarray(st["done"]), 9, env._epoch - 1, st["ball_xy"], None, st["step_count"],
                    st["reward"], st["done"], None, env_base=N - 4096)
        O.step_v0(p, lay, a_tail, st["ball_xy"], st["step_count"], st["reward"], st["done"], st["goal_count"], ref)
    assert (_np(env.obs[tail]) == ref).all() and (_np(env.ball_xy[tail]) == st["ball_xy"]).all()
    bits = torch.from_numpy(np.vectorize(lambda c: {ord("W"): 2, ord("X"): 4, ord("B"): 8}.get(c, 0))(lay).astype(np.int32)).cuda()
    for lo in range(0, N, 1 << 21):                                  # in slices: the checks allocate temporaries
        sl = slice(lo, lo + (1 << 21))
        obs = env.obs[sl]
        ball = obs & 1
        assert bool((ball.flatten(1).sum(dim=1) == 1).all())
        idx = env.ball_xy[sl, 0].long() * G + env.ball_xy[sl, 1].long()
        assert bool((ball.flatten(1).gather(1, idx[:, None]) == 1).all())
        assert bool(((obs & ~1) == bits).all())
    out = env.planes()                                               # 16.2 GB of float planes: the x1 render past 2^32 bytes
    want = ((env.obs[tail][:, None] & torch.tensor([1, 2, 4, 8], dtype=torch.int32, device="cuda")[None, :, None, None]) != 0)
    assert torch.equal(out[tail], want.to(torch.float32))
    assert torch.equal(out[:4096], ((env.obs[:4096][:, None] & torch.tensor([1, 2, 4, 8], dtype=torch.int32,
                                                                           device="cuda")[None, :, None, None]) != 0).to(torch.float32))


def test_four_million_per_env_32x32_tail_against_the_oracle():
    """17 GB of planes and 4 GB of per-env layouts on one GPU: the last and the first 1 024 envs against the
    oracle (per-env kernel, one wave per env: the env index times 4 096 bytes passes 2^32 at env 2^20)."""
    N, G, K, T = 1 << 22, 32, 1024, 4
    gen = torch.Generator(device="cuda").manual_seed(11)
    lay = torch.where(torch.rand((N, G, G), device="cuda", generator=gen) < 0.25, ord("W"), ord("B")).to(torch.uint8)
    lay[:, 0, :] = ord("W"); lay[:, -1, :] = ord("W"); lay[:, :, 0] = ord("W"); lay[:, :, -1] = ord("W")
    lay[:, 1, 1] = ord("S")
    lay[:, G - 2, G - 2] = ord("X")
    env = PKG.LmazeVecEnv(N, variant="v0", per_env_layouts=lay, seed=2, validate=False)
    p = O.params(O.VARIANT_V0, G, O.LAYOUT_PER_ENV)
    h = env.host_state()
    mirrors = []
    for sl in (slice(0, K), slice(N - K, N)):
        st = {k: np.array(v[sl], copy=True) for k, v in h.items()}
        mirrors.append((sl, st, np.ascontiguousarray(_np(lay[sl])), np.zeros((K, G, G), np.int32)))
    for t in range(T):
        a = torch.randint(0, 4, (N,), dtype=torch.int32, device="cuda", generator=gen)
        env.step(a)
        for sl, st, lay_np, ref in mirrors:
            O.step_v0(p, lay_np, np.ascontiguousarray(_np(a[sl])), st["ball_xy"], st["step_count"], st["reward"],
                      st["done"], st["goal_count"], ref)
    for sl, st, lay_np, ref in mirrors:
        assert (_np(env.obs[sl]) == ref).all() and (_np(env.ball_xy[sl]) == st["ball_xy"]).all(), sl


def test_online_tuner_picks_a_policy_without_changing_results():
    """Large shared-layout batches time the launch policies on their own first steps: the run with the tuner
    cycling through its candidates and the run with a fixed policy produce the same bits, and after enough
    launches the tuner has chosen and stepped aside."""
    N, G = 1 << 19, 11                     # 254 MB of obs: the streaming regime
    lay = L.to_codes(L.open_room(G, (5, 5)))
    tuned = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=6, online_autotune=True)
    fixed = PKG.LmazeVecEnv(N, variant="v0", layout=lay, seed=6)          # opt-in: off unless asked for
    small = PKG.LmazeVecEnv(4096, variant="v0", layout=lay, seed=6, online_autotune=True)
    assert tuned._tuner is not None and fixed._tuner is None and small._tuner is None
    assert tuned.tuning_progress() == (0, 12 * len(tuned.ONLINE_CANDIDATES)) and fixed.tuning_progress() is None
    acts = torch.randint(0, 4, (8, N), dtype=torch.int32, device="cuda")
    need = tuned._tuner.warm + tuned._tuner.samples * len(tuned.ONLINE_CANDIDATES) + 64
    for t in range(need):
        tuned.step(acts[t % 8], auto_reset=True)
        fixed.step(acts[t % 8], auto_reset=True)
        if t % 97 == 0:
            assert (tuned.obs == fixed.obs).all() and (tuned._state == fixed._state).all(), t
    torch.cuda.synchronize()
    for t in range(16):                    # the last pairs are collected on later calls
        tuned.step(acts[t % 8], auto_reset=True)
        fixed.step(acts[t % 8], auto_reset=True)
    assert tuned._tuner is None and tuned.tuned_policy in tuned.ONLINE_CANDIDATES
    assert tuned.params.launch_hint == tuned.launch_hint_of(*tuned.tuned_policy)
    assert (tuned.obs == fixed.obs).all() and (tuned._state == fixed._state).all()
    tuned.set_launch_policy(8, 1)
    assert tuned.params.launch_hint == 0x18
