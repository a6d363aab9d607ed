"""CPU property tests (hypothesis): the oracle's batched loops equal N independent single-env calls,
per-env layouts equal the shared layout when every env carries the same maze, and the xE render is the
exact nearest-neighbour replication -- the invariants the GPU parity tests lean on."""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle_lib as O
from helpers import bordered_random_layouts, random_free_cells


@settings(max_examples=25, deadline=None)
@given(G=st.integers(4, 20), n=st.integers(1, 40), seed=st.integers(0, 10 ** 6), variant=st.sampled_from([0, 3]),
       steps=st.integers(1, 12))
def test_batched_equals_independent_envs(G, n, seed, variant, steps):
    rs = np.random.RandomState(seed)
    lay = bordered_random_layouts(n, G, seed)
    ball = random_free_cells(lay, seed + 1, forbid=(ord("W"),) if variant == 3 else (ord("W"), ord("X")))
    goal = random_free_cells(lay, seed + 2, forbid=(ord("W"),))
    p = O.params(variant, G, O.LAYOUT_PER_ENV)
    acts = rs.randint(-1, 6, (steps, n)).astype(np.int32)

    def run(lo, hi):
        m = hi - lo
        b, g = ball[lo:hi].copy(), goal[lo:hi].copy()
        sc, rw, dn, gc = np.zeros(m, np.int32), np.zeros(m, np.float32), np.zeros(m, np.uint8), np.zeros(m, np.int32)
        obs = np.zeros((m, G, G), np.int32)
        l = np.ascontiguousarray(lay[lo:hi])
        for t in range(steps):
            a = np.ascontiguousarray(acts[t, lo:hi])
            if variant == 3:
                O.step_v3(p, l, a, b, g, sc, rw, dn, obs)
            else:
                O.step_v0(p, l, a, b, sc, rw, dn, gc, obs)
        return b, sc, rw.view(np.uint32), dn, gc, obs

    whole = run(0, n)
    for i in range(n):
        one = run(i, i + 1)
        for w, o in zip(whole, one):
            assert (w[i:i + 1] == o).all()


@settings(max_examples=20, deadline=None)
@given(G=st.integers(4, 16), n=st.integers(1, 30), seed=st.integers(0, 10 ** 6))
def test_per_env_layouts_equal_shared_when_identical(G, n, seed):
    rs = np.random.RandomState(seed)
    one = bordered_random_layouts(1, G, seed)
    lay = np.repeat(one, n, axis=0)
    ball = random_free_cells(lay, seed + 1)
    a = rs.randint(0, 4, n).astype(np.int32)
    outs = []
    for mode, l in ((O.LAYOUT_SHARED, np.ascontiguousarray(one[0])), (O.LAYOUT_PER_ENV, np.ascontiguousarray(lay))):
        b = ball.copy()
        sc, rw, dn, gc = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.int32)
        obs = np.zeros((n, G, G), np.int32)
        O.step_v0(O.params(0, G, mode), l, a, b, sc, rw, dn, gc, obs)
        outs.append((b, rw.view(np.uint32), dn, obs))
    for x, y in zip(*outs):
        assert (x == y).all()


@settings(max_examples=20, deadline=None)
@given(G=st.integers(1, 12), E=st.integers(1, 8), C=st.integers(1, 8), seed=st.integers(0, 10 ** 6))
def test_expanded_render_is_exact_replication(G, E, C, seed):
    rs = np.random.RandomState(seed)
    obs = rs.randint(0, 256, (3, G, G)).astype(np.int32)
    masks = [int(m) for m in rs.randint(1, 256, C)]
    out = O.render_expanded(obs, G, E, masks)
    want = np.stack([np.repeat(np.repeat(((obs & m) != 0).astype(np.float32), E, axis=1), E, axis=2) for m in masks], axis=1)
    assert out.shape == want.shape and (out == want).all()
