"""TEST INFRASTRUCTURE ONLY (oracle/): NumPy-vectorised restatement of the v0 step.

The third CPU leg SURVEY.md 8(d) asks for beside the scalar C restatement (1 thread) and the threaded one:
what a NumPy user would write to batch the reference's step() over N mazes.  Checked bit for bit against
the C oracle and the reference's recorded rollouts (tests/test_oracle_golden.py); only tests/ and bench.py's
cpu_baseline leg import it.

Reference: gym_lmaze/envs/lmaze_env.py:146-249 (step, isEpisodeFinished), planes lmaze_env.py:92-107,208-215.
"""
import numpy as np

OBS_BALL, OBS_WALL, OBS_GOAL, OBS_FREE = 1, 2, 4, 8
W, B, X = ord("W"), ord("B"), ord("X")


def static_bits(layout):
    """int32 planes of the layout without the ball: wall 'W', goal 'X', blank 'B' ('S' is in no static plane;
    lmaze_env.py:92-107)."""
    lay = np.asarray(layout, dtype=np.uint8)
    return (np.where(lay == W, OBS_WALL, 0) | np.where(lay == X, OBS_GOAL, 0) | np.where(lay == B, OBS_FREE, 0)).astype(np.int32)


# action id -> (row delta, column delta); anything else stays put (lmaze_env.py:153-170)
_OX = np.array([-1, 1, 0, 0], np.int32)
_OY = np.array([0, 0, -1, 1], np.int32)


def step_v0(layout, static, action, ball_xy, step_count, reward, done, goal_count, obs,
            step_limit=100, reward_wall=-1.0, reward_move=-0.01, reward_goal=100.0):
    """One v0 step of every env, in place.  layout uint8[G,G] (shared) or uint8[N,G,G]; static =
    static_bits(layout); the state arrays are those of the C oracle (oracle_lib.step_v0)."""
    n = action.shape[0]
    G = layout.shape[-1]
    idx = np.arange(n)
    valid = (action >= 0) & (action < 4)
    a = np.where(valid, action, 0)
    ox = np.where(valid, _OX[a], 0)
    oy = np.where(valid, _OY[a], 0)
    step_count += 1                                              # lmaze_env.py:151
    tx = np.clip(ball_xy[:, 0] + ox, 0, G - 1)
    ty = np.clip(ball_xy[:, 1] + oy, 0, G - 1)
    c = layout[tx, ty] if layout.ndim == 2 else layout[idx, tx, ty]   # lmaze_env.py:172
    wall, blank, goal = c == W, c == B, c == X
    move = blank | goal                                          # lmaze_env.py:176-193
    ball_xy[:, 0] = np.where(move, tx, ball_xy[:, 0])
    ball_xy[:, 1] = np.where(move, ty, ball_xy[:, 1])
    # no else branch: a target that is none of W/B/X ('S') leaves the reward of the previous step
    r = np.where(wall, np.float32(reward_wall), np.where(blank, np.float32(reward_move),
                                                         np.where(goal, np.float32(reward_goal), reward)))
    reward[:] = r.astype(np.float32)
    if goal_count is not None:
        goal_count += goal.astype(np.int32)                      # lmaze_env.py:195
    done[:] = ((reward == np.float32(reward_goal)) | (step_count == step_limit)).astype(np.uint8)   # lmaze_env.py:246-249
    if obs is not None:                                          # lmaze_env.py:208-215, compact form
        obs[:] = static if static.ndim == 3 else static[None]
        obs[idx, ball_xy[:, 0], ball_xy[:, 1]] |= OBS_BALL
