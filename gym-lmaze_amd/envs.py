"""Drop-in env classes: same names, constructor defaults, attributes and return tuples as
the reference's gym.Env subclasses, backed by the HIP step path.

  LmazeEnv      <- gym_lmaze/envs/lmaze_env.py:11-257      (id lmaze-v0)
  LmazeEnv_v3   <- gym_lmaze/envs/lmaze_env_v3.py:17-402   (id lmaze-v3)

`Class()` with no arguments is one env and behaves like the reference object: numpy
float32 observations in the reference layout, Python float reward, bool done, and the
action echoed in the 4th slot (lmaze_env.py:237).  `Class(num_envs=N)` is the batched
extension: torch tensors on the GPU, compact int32 planes unless obs_mode="expanded".

What stays on the host, by design: the single-env reset() placement, which consumes
Python's global `random` in the reference's own draw order (lmaze_env.py:70-78) so a
seeded `random` gives the same start cells as the reference; and type conversion of the
returned scalars.  Everything step() computes runs in the kernel.
"""
import random

import numpy as np

from . import layouts as L
from ._staging import HostStaging
from .compat import Box, Discrete, Env
from .vec_env import LmazeVecEnv

_NOOP_ACTION = -1


def _f32_bits(x):
    return int(np.float32(x).view(np.uint32))


class _LmazeBase(Env):
    metadata = {'render.modes': ['human']}
    _variant = "v0"

    # ------------------------------------------------------------------ construction
    def _setup(self, num_envs, device, layout, per_env_layouts, obs_mode, seed, expansion):
        self.num_envs = int(num_envs)
        self._device = device
        self._seed = 0 if seed is None else int(seed)
        self._per_env_layouts = per_env_layouts
        self._single = self.num_envs == 1 and per_env_layouts is None
        self.obs_mode = obs_mode or ("expanded" if self._single else "compact")
        if self.obs_mode not in ("expanded", "compact"):
            raise ValueError("obs_mode must be 'expanded' or 'compact'")
        self._grid_chars = L.to_char_grid(layout) if layout is not None else None
        self._expansion = expansion
        self._core = None
        self._stage = None
        self._dirty = True
        self._host = None

    def _build_core(self):
        kw = dict(variant=self._variant, device=self._device, expansion=self.expansionRatio,
                  step_limit=self._step_limit(), rewards=(self.negativeNominal, self.positiveNominal,
                                                          self.positiveFull), seed=self._seed)
        if self._per_env_layouts is not None:
            self._core = LmazeVecEnv(self.num_envs, per_env_layouts=self._per_env_layouts, **kw)
        else:
            self._core = LmazeVecEnv(self.num_envs, layout=self._grid_chars, **kw)
        self._stage = HostStaging(self._core.device) if self._single else None
        self._dirty = False

    def _step_limit(self):
        return 100

    # `grid`, `realgrid`, `gridsize` are read on every reset()/step() in the reference
    # (lmaze_env.py:67-141), so overriding them re-sizes the env at the next reset().
    @property
    def grid(self):
        return self._grid_chars

    @grid.setter
    def grid(self, value):
        self._grid_chars = L.to_char_grid(value)
        self._dirty = True

    @property
    def realgrid(self):
        return self._grid_chars.shape[0]

    @realgrid.setter
    def realgrid(self, value):
        if int(value) != self._grid_chars.shape[0]:
            raise ValueError("realgrid must equal the side of `grid` (set env.grid first)")

    @property
    def gridsize(self):
        return self.realgrid * self.expansionRatio

    @gridsize.setter
    def gridsize(self, value):
        pass  # derived; accepted so reference-style attribute overrides do not fail

    # ------------------------------------------------------------------ host mirror (single env)
    def _sync_host(self):
        if self._stage is not None:
            self._host = self._core.host_state(raw=self._stage.fetch(state=self._core._state)["state"].copy())
        else:
            self._host = self._core.host_state()
        return self._host

    def _step_single(self, action):
        """N = 1: upload the action, step, render xE, and bring observation + scalars back with one sync.
        Returns (page-locked view of the (C,GE,GE) observation or None in compact mode, host scalars)."""
        core, stage = self._core, self._stage
        expanded = self.obs_mode == "expanded"

        def body(act_dev):
            core.step(act_dev)
            return {"obs": core.expanded()[0], "state": core._state} if expanded else {"state": core._state}

        got = stage.step(("step", expanded), action, body)       # one hipGraph launch + one sync
        self._host = core.host_state(raw=got["state"].copy())
        return got.get("obs"), self._host

    def _scalar(self, key, idx=None):
        h = self._host if self._host is not None else self._sync_host()
        v = h[key][0]
        return v if idx is None else v[idx]

    def _reward_to_python(self, r32):
        """float32 from the device -> the double the reference would have returned."""
        table = getattr(self, "_reward_table", None)
        if table is None:
            table = {_f32_bits(v): float(v) for v in (self.negativeNominal, self.positiveNominal,
                                                      self.positiveFull, -0.0, 0.0)}
            table[_f32_bits(-0.0)] = -0.0
            self._reward_table = table
        return table.get(int(np.float32(r32).view(np.uint32)), float(r32))

    ball_x0 = property(lambda s: int(s._scalar("ball_xy", 0)), lambda s, v: s._poke_ball(x=v))
    ball_y0 = property(lambda s: int(s._scalar("ball_xy", 1)), lambda s, v: s._poke_ball(y=v))
    goal_x = property(lambda s: int(s._scalar("goal_xy", 0)), lambda s, v: s._poke_goal(x=v))
    goal_y = property(lambda s: int(s._scalar("goal_xy", 1)), lambda s, v: s._poke_goal(y=v))

    def _poke_ball(self, x=None, y=None):
        b = np.array(self._sync_host()["ball_xy"], copy=True)
        if x is not None:
            b[:, 0] = int(x)
        if y is not None:
            b[:, 1] = int(y)
        self._core.set_state(ball_xy=b)
        self._host = None

    def _poke_goal(self, x=None, y=None):
        g = np.array(self._sync_host()["goal_xy"], copy=True)
        if x is not None:
            g[:, 0] = int(x)
        if y is not None:
            g[:, 1] = int(y)
        self._core.set_state(goal_xy=g)
        self._host = None

    @property
    def stepCount(self):
        return int(self._scalar("step_count")) if self._single else self._core.step_count

    @stepCount.setter
    def stepCount(self, v):
        self._core.set_state(step_count=np.full(self.num_envs, int(v), np.int32))
        self._host = None

    # ------------------------------------------------------------------ observations
    def _obs_out(self, fresh):
        """Observation in the configured mode; single env -> numpy in the reference layout."""
        if self.obs_mode == "compact":
            return self._core.obs
        full = self._core.expanded()
        if not self._single:
            return full
        return self._stage.fetch(obs=full[0])["obs"].copy()  # a fresh array per call, like lmaze_env.py:217

    @property
    def core(self):
        """The batched engine (LmazeVecEnv): state tensors, obs buffer, set_state()."""
        return self._core

    # ------------------------------------------------------------------ reference extras
    def render(self, mode='human', close=False):
        # lmaze_env.py:55-59 only flips the flag; the cv2 window itself is out of scope
        self.VISUALIZE = (mode == 'human')

    def rendering(self, msg):
        self.VISUALIZE = msg

    def writing(self, msg):
        self.SAVEFRAME = msg


class LmazeEnv(_LmazeBase):
    """lmaze-v0: full view, 4 actions, obs (4, 7G, 7G) -- lmaze_env.py:11-257."""
    _variant = "v0"

    def __init__(self, num_envs=1, device=None, layout=None, per_env_layouts=None, obs_mode=None, seed=None):
        print("init-init")                                   # lmaze_env.py:15
        self.action_space = Discrete(4)                      # lmaze_env.py:16
        self.expansionRatio = 7                              # lmaze_env.py:18
        self.negativeNominal = -1.0                          # lmaze_env.py:21
        self.positiveNominal = -0.01                         # lmaze_env.py:22
        self.positiveFull = 100.0                            # lmaze_env.py:23
        self.RANDOM_BALL = True                              # lmaze_env.py:25
        self.VISUALIZE = False                               # lmaze_env.py:26
        self._setup(num_envs, device, layout if layout is not None else L.V0_GRID_12, per_env_layouts,
                    obs_mode, seed, 7)
        if per_env_layouts is not None:
            g = int(per_env_layouts.shape[-1])
            self._grid_chars = np.full((g, g), "W")          # placeholder: layouts live on the device
        self.observation_space = Box(0.0, 1.0, shape=(4, self.gridsize, self.gridsize))  # lmaze_env.py:20
        self.reset()                                         # lmaze_env.py:52
        print("init-end")                                    # lmaze_env.py:53

    # goalCount lives on the device (lmaze_env.py:24,195)
    @property
    def goalCount(self):
        return int(self._scalar("goal_count")) if self._single else self._core.goal_count

    @goalCount.setter
    def goalCount(self, v):
        if self._core is not None:
            self._core.set_state(goal_count=np.full(self.num_envs, int(v), np.int32))
            self._host = None

    @property
    def reward(self):
        return self._reward_to_python(self._scalar("reward")) if self._single else self._core.reward

    @property
    def state(self):
        """flat float32 [4*G*G] plane stack of env 0 (lmaze_env.py:50,67)."""
        o = self._core.obs[0].cpu().numpy()
        return np.concatenate([((o & m) != 0).astype(np.float32).reshape(-1) for m in self._core.channel_mask])

    def reset(self, mask=None):
        if self._dirty or self._core is None:
            self._build_core()
            self.observation_space = Box(0.0, 1.0, shape=(4, self.gridsize, self.gridsize))
        core = self._core
        if self._single:
            grid = self._grid_chars
            if self.RANDOM_BALL:                             # lmaze_env.py:70-78, same draw order
                x = y = 0
                while grid[x][y] == 'W' or grid[x][y] == 'X':
                    x = random.randint(1, self.realgrid - 2)
                    y = random.randint(1, self.realgrid - 2)
            else:                                            # lmaze_env.py:87-89
                s = np.where(grid == 'S')
                x, y = int(s[0][0]), int(s[1][0])
            core.set_state(ball_xy=np.array([[x, y]], np.int32), step_count=np.zeros(1, np.int32),
                           reward=np.array([-0.0], np.float32), done=np.zeros(1, np.uint8))
            core.observe()
        else:
            core.reset(mask)
        self._host = None
        return self._obs_out(fresh=True)

    def step(self, msg):
        core = self._core
        if self._single:
            a = int(msg)                                     # lmaze_env.py:148
            view, h = self._step_single(a if -2 ** 31 <= a < 2 ** 31 else _NOOP_ACTION)
            obs = view.copy() if view is not None else core.obs      # a fresh array per call (lmaze_env.py:217)
            return obs, self._reward_to_python(h["reward"][0]), bool(h["done"][0]), a
        core.step(msg)
        self._host = None
        return self._obs_out(fresh=False), core.reward, core.done, msg

    def initState(self):
        return self.state, self.reward, self.isEpisodeFinished(), {'newState': True}

    def isEpisodeFinished(self):                             # lmaze_env.py:246-249
        if self._single:
            return bool(self.reward == self.positiveFull or self.stepCount == 100)
        return (self._core.reward == self.positiveFull) | (self._core.step_count == 100)


def _decode_v3_action(goal):
    """lmaze_env_v3.py:236-247: only these strings move; anything else (ints included) is a no-op."""
    if isinstance(goal, str):
        if goal == "left" or goal == "0":
            return 0
        if goal == "right" or goal == "1":
            return 1
        if goal == "up" or goal == "2":
            return 2
        if goal == "down" or goal == "3":
            return 3
    return _NOOP_ACTION


class LmazeEnv_v3(_LmazeBase):
    """lmaze-v3: full view 18x18, string actions, random goal, obs (3, 4G, 4G) -- lmaze_env_v3.py:17-402."""
    _variant = "v3"

    def __init__(self, num_envs=1, device=None, layout=None, per_env_layouts=None, obs_mode=None, seed=None):
        self.state_type = "fullView"                         # lmaze_env_v3.py:24
        self.expansionRatio = 4                              # lmaze_env_v3.py:76
        self.channel = 3                                     # lmaze_env_v3.py:89-90
        self.stateChannel = 3
        self.step_limit = 100                                # lmaze_env_v3.py:96
        self.negativeNominal = -1.0
        self.positiveNominal = -0.01
        self.positiveFull = 100.0
        self.RANDOM_BALL = True                              # lmaze_env_v3.py:100-104
        self.RANDOM_GOAL = True
        self.AUTO_VISUALIZE = False
        self.SAVEFRAME = False
        self.VISUALIZE = False
        self.localDone = False
        self.fovealStepCount = 0
        self.dir = "."
        self._setup(num_envs, device, layout if layout is not None else L.V3_GRID_18, per_env_layouts,
                    obs_mode, seed, 4)
        if per_env_layouts is not None:
            g = int(per_env_layouts.shape[-1])
            self._grid_chars = np.full((g, g), "W")
        self.observation_space = Box(0.0, 1.0, shape=(self.stateChannel, self.gridsize, self.gridsize))
        self.action_space = Discrete(4)                      # lmaze_env_v3.py:93
        self.retStateExpanded = None
        self.reset()                                         # lmaze_env_v3.py:125

    def _step_limit(self):
        return self.step_limit

    @property
    def fovea(self):                                         # lmaze_env_v3.py:77 (== realgrid)
        return self.realgrid

    @property
    def originalReward(self):
        return self._reward_to_python(self._scalar("reward")) if self._single else self._core.reward

    @property
    def state(self):
        """float32 [3,G,G] plane stack of env 0 (lmaze_env_v3.py:142,166-167)."""
        o = self._core.obs[0].cpu().numpy()
        return np.stack([((o & m) != 0).astype(np.float32) for m in self._core.channel_mask])

    def _single_obs(self, arr=None):
        """v3 hands out one reused buffer (lmaze_env_v3.py:122,206,400)."""
        if arr is None:
            arr = self._stage.fetch(obs=self._core.expanded()[0])["obs"]
        if self.retStateExpanded is None or self.retStateExpanded.shape != arr.shape:
            self.retStateExpanded = np.zeros(arr.shape, dtype=np.float32)
        np.copyto(self.retStateExpanded, arr)
        return self.retStateExpanded

    def reset(self, mode="train", mask=None):
        if self._dirty or self._core is None:
            self._build_core()
            self.observation_space = Box(0.0, 1.0, shape=(self.stateChannel, self.gridsize, self.gridsize))
            self.retStateExpanded = None
        core = self._core
        if self._single:
            grid, G = self._grid_chars, self.realgrid
            h = self._sync_host()
            gx, gy = int(h["goal_xy"][0][0]), int(h["goal_xy"][0][1])
            bx, by = int(h["ball_xy"][0][0]), int(h["ball_xy"][0][1])
            if mode == "test":                               # lmaze_env_v3.py:145-146
                gx, gy = 8, 8
            elif self.RANDOM_GOAL:                           # lmaze_env_v3.py:147-152
                x, y = 0, 0
                while grid[x][y] == 'W':
                    x = random.randint(1, G - 2)
                    y = random.randint(1, G - 2)
                gx, gy = x, y
            if mode == "test":                               # lmaze_env_v3.py:154-155
                bx, by = 7, 8
            elif self.RANDOM_BALL:                           # lmaze_env_v3.py:156-161
                x, y = 0, 0
                while grid[x][y] == 'W' or (x == gx and y == gy):
                    x = random.randint(1, G - 2)
                    y = random.randint(1, G - 2)
                bx, by = x, y
            else:                                            # lmaze_env_v3.py:163-164
                s = np.where(grid == 'S')
                bx, by = int(s[0][0]), int(s[1][0])
            core.set_state(ball_xy=np.array([[bx, by]], np.int32), goal_xy=np.array([[gx, gy]], np.int32),
                           step_count=np.zeros(1, np.int32), reward=np.array([-0.0], np.float32),
                           done=np.zeros(1, np.uint8))
            core.observe()
            self._host = None
            return self._single_obs() if self.obs_mode == "expanded" else core.obs
        core.reset(mask)
        self._host = None
        return self._obs_out(fresh=False)

    def step(self, goal):
        core = self._core
        if self._single:
            view, h = self._step_single(_decode_v3_action(goal))
            obs = self._single_obs(view) if view is not None else core.obs
            return obs, self._reward_to_python(h["reward"][0]), bool(h["done"][0]), goal
        core.step(goal)   # batched: int32 ids, 0..3 move, anything else is the no-op
        self._host = None
        return self._obs_out(fresh=False), core.reward, core.done, goal
