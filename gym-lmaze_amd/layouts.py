"""Maze layout tables: the reference's live layouts as row strings (data, not code).

Cell characters are the reference's own: 'W' wall, 'B' blank, 'S' start, 'X' goal marker
(gym_lmaze/envs/lmaze_env.py:37-48).  tests/test_layouts.py checks every table against the
layout bytes stored in the golden fixtures (which were read off the reference objects).
"""
import numpy as np

# lmaze_env.py:37-48 (lmaze-v0)
V0_GRID_12 = (
    "WWWWWWWWWWWW",
    "WSBBBBBBBBBW",
    "WBWWBWWWWWBW",
    "WBBBWBBBBBBW",
    "WBBWBBBWWWBW",
    "WBWBWXWBBBBW",
    "WBBBWBBWWBBW",
    "WBWBWBBBBWBW",
    "WBWBBBWBBBBW",
    "WBBWWWBBWWBW",
    "WBBBBBBBBBBW",
    "WWWWWWWWWWWW",
)

# lmaze_env_v1.py:40-53 (lmaze-v1), 2-cell 'W' pad for the 5x5 fovea
V1_GRID_14 = (
    "WWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWW",
    "WWSBBBWBBBBWWW",
    "WWBWWWWBWWBWWW",
    "WWBWBWBBBWBWWW",
    "WWBWBWBWBWBWWW",
    "WWBBBWXWBWWWWW",
    "WWBWBWBWBWBWWW",
    "WWBWBWBWBWBWWW",
    "WWBWBBBWBWBWWW",
    "WWBWWWWWBWBWWW",
    "WWBBBBWBBBBWWW",
    "WWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWW",
)

# lmaze_env_v3.py:26-43 (lmaze-v3), 4-cell pad
V3_GRID_18 = (
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWSBBBWBBBBBWWWW",
    "WWWWBWWWWBWWBBWWWW",
    "WWWWBWBBBBBWBBWWWW",
    "WWWWBWBBBBBWBBWWWW",
    "WWWWBBBBXBBWWWWWWW",
    "WWWWBWBBBBBWBBWWWW",
    "WWWWBWBBBBBWBBWWWW",
    "WWWWBWBBBBBWBBWWWW",
    "WWWWBWWWWWBWBBWWWW",
    "WWWWBBBBWBBBBBWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
)

# lmaze_env_v2.py:309-405 branch random == 1 (same table in v4/v5/v6)
FOVEAL_GRID_18_1 = (
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWSBBBWBBBBBWWWW",
    "WWWWBWWWWBWWBBWWWW",
    "WWWWBWBWBBBWBBWWWW",
    "WWWWBWBWBWBWBBWWWW",
    "WWWWBBBWXWBWWWWWWW",
    "WWWWBWBWBWBWBBWWWW",
    "WWWWBWBWBWBWBBWWWW",
    "WWWWBWBBBWBWBBWWWW",
    "WWWWBWWWWWBWBBWWWW",
    "WWWWBBBBWBBBBBWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
)

# lmaze_env_v2.py:309-405 branch random == 2 (same table in v4/v5/v6)
FOVEAL_GRID_18_2 = (
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWSBBBBBBBBBWWWW",
    "WWWWBWWWBWWWWBWWWW",
    "WWWWBWBBBBBWWBWWWW",
    "WWWWBWBWBWBWWBWWWW",
    "WWWWBWBWXWBWWBWWWW",
    "WWWWBWBWWWBWWBWWWW",
    "WWWWBWBWWWBBBBWWWW",
    "WWWWBWBBBBBWWBWWWW",
    "WWWWBWWWBWWWWBWWWW",
    "WWWWBBBBBBBBBBWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
)

# lmaze_env_v2.py:309-405 branch random == 3 (same table in v4/v5/v6)
FOVEAL_GRID_18_3 = (
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWSBBBBBBBBBWWWW",
    "WWWWBWBWWWWWBBWWWW",
    "WWWWBWBBBWBBBBWWWW",
    "WWWWBWBBBWBWBBWWWW",
    "WWWWBWBWXWBWBBWWWW",
    "WWWWBWBWBBBWBBWWWW",
    "WWWWBWBWBWBWBBWWWW",
    "WWWWBBBWBBBBBBWWWW",
    "WWWWBWWWWWWWBBWWWW",
    "WWWWBBBBBBBBBBWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
)

# lmaze_env_v2.py:309-405 branch random == 4 (same table in v4/v5/v6)
FOVEAL_GRID_18_4 = (
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWSBBBBBBBBBWWWW",
    "WWWWBBWWWBWWBBWWWW",
    "WWWWBBBWBBBWBBWWWW",
    "WWWWBWBWBWBWBBWWWW",
    "WWWWBWBWXWBWBBWWWW",
    "WWWWBWBWBWBWBBWWWW",
    "WWWWBWBWBWBBBBWWWW",
    "WWWWBWBBBWBWBBWWWW",
    "WWWWBWBWWWWWBBWWWW",
    "WWWWBBBBBBBBBBWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
)

# lmaze_env_v2.py:309-405 branch random == 5 (same table in v4/v5/v6)
FOVEAL_GRID_18_5 = (
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWSBBBBBBBBBWWWW",
    "WWWWBWWWWBWWBBWWWW",
    "WWWWBBWBBBBWBBWWWW",
    "WWWWBBBBWWBWBBWWWW",
    "WWWWBWWXWWWWBBWWWW",
    "WWWWBWWBWWBWBBWWWW",
    "WWWWBWWBWWBWBBWWWW",
    "WWWWBWBBBBBWBBWWWW",
    "WWWWBWBWWBWWBBWWWW",
    "WWWWBBBBBBBBBBWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
    "WWWWWWWWWWWWWWWWWW",
)

FOVEAL_GRIDS_18 = (FOVEAL_GRID_18_1, FOVEAL_GRID_18_2, FOVEAL_GRID_18_3, FOVEAL_GRID_18_4, FOVEAL_GRID_18_5)


def to_char_grid(rows):
    """rows of characters -> numpy array of 1-char strings, the type the reference keeps in `self.grid`."""
    if isinstance(rows, np.ndarray) and rows.dtype.kind in "US":
        return rows
    return np.array([list(r) for r in rows])


def to_codes(grid):
    """char grid / row strings / uint8 codes -> contiguous uint8[G,G] of ASCII codes."""
    if isinstance(grid, np.ndarray) and grid.dtype == np.uint8:
        codes = grid
    else:
        g = to_char_grid(grid)
        codes = np.frombuffer("".join("".join(r) for r in g).encode("ascii"), dtype=np.uint8).reshape(g.shape)
    if codes.ndim < 2 or codes.shape[-1] != codes.shape[-2]:
        raise ValueError("layouts must be square [.., G, G]; got %r" % (codes.shape,))
    return np.ascontiguousarray(codes)


def validate(codes, need_goal_marker=True):
    """The step path never range-checks (neither does the reference): require the full 'W'
    border that makes every reachable index valid, plus the markers reset() looks up."""
    c = np.asarray(codes)
    W = ord("W")
    if not ((c[..., 0, :] == W).all() and (c[..., -1, :] == W).all()
            and (c[..., :, 0] == W).all() and (c[..., :, -1] == W).all()):
        raise ValueError("every layout needs a full 'W' border (the reference would raise IndexError "
                         "or wrap to the opposite edge when the ball walks off an open border)")
    if not np.isin(c, [ord(ch) for ch in "WBSX"]).all():
        raise ValueError("layout cells must be one of 'W', 'B', 'S', 'X'")
    if need_goal_marker and not (c == ord("X")).reshape(c.shape[:-2] + (-1,)).any(-1).all():
        raise ValueError("layout has no 'X' cell (reset() looks the goal up with np.where, lmaze_env.py:100-102)")


def open_room(G, goal=None):
    """G x G room: 'W' border, 'S' at (1,1), 'X' at `goal` (default centre) -- SURVEY 8(d) C3 layout."""
    g = np.full((G, G), "B")
    g[0, :] = g[-1, :] = g[:, 0] = g[:, -1] = "W"
    g[1, 1] = "S"
    gx, gy = goal if goal is not None else (G // 2, G // 2)
    g[gx, gy] = "X"
    return g


# the commented 8x8 literal of lmaze_env.py:28-35 with the open last row/column closed by 'W'
GRID_8_BORDERED = (
    "WWWWWWWW",
    "WSBBBWWW",
    "WBWWWWWW",
    "WBWBWBBW",
    "WBWBWBWW",
    "WBBBWXWW",
    "WBWBWBWW",
    "WWWWWWWW",
)


def random_walled(num_envs, G, device, p_wall=0.25, seed=7, block=1 << 17):
    """Per-env random mazes of SURVEY 8(d) C5 as uint8[N,G,G] on `device`: a 'W' border, interior cells walls with
    probability p_wall (i.i.d., torch's device generator is Philox, seed `seed`), 'X' on a uniformly chosen free cell of
    each maze; every other free cell 'B'.  The ball goes to another free cell at reset() (lmaze_env.py:70-78: uniform
    over the cells that are neither 'W' nor 'X').  Built in blocks of `block` envs to bound the scratch memory."""
    import torch
    gen = torch.Generator(device=device).manual_seed(int(seed))
    out = torch.empty((num_envs, G, G), dtype=torch.uint8, device=device)
    for lo in range(0, num_envs, block):
        n = min(block, num_envs - lo)
        lay = torch.where(torch.rand((n, G, G), device=device, generator=gen) < p_wall, ord("W"), ord("B")).to(torch.uint8)
        lay[:, 0, :] = ord("W"); lay[:, -1, :] = ord("W"); lay[:, :, 0] = ord("W"); lay[:, :, -1] = ord("W")
        score = torch.rand((n, G * G), device=device, generator=gen)
        score = torch.where(lay.view(n, -1) == ord("B"), score, torch.full_like(score, -1.0))
        pick = score.argmax(dim=1)
        has_free = score.gather(1, pick[:, None])[:, 0] >= 0            # a maze without a free cell keeps no 'X'
        flat = lay.view(n, -1)
        rows = torch.nonzero(has_free)[:, 0]
        flat[rows, pick[rows]] = ord("X")
        out[lo:lo + n] = lay
    return out
