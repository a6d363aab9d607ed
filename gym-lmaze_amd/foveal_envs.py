"""Drop-in classes of the foveal variants, backed by LmazeFovealVecEnv (HIP step path).

  LmazeEnv_v1  <- gym_lmaze/envs/lmaze_env_v1.py:14-324   (id lmaze-v1)
  LmazeEnv_v2  <- gym_lmaze/envs/lmaze_env_v2.py:17-435   (id lmaze-v2)
  LmazeEnv_v4  <- gym_lmaze/envs/lmaze_env_v4.py:17-482   (id lmaze-v4)

`Class()` is one env with the reference's return tuples (v1: 6-tuple, v1:200; v2/v4: 4-tuple with
the action in the last slot and ONE reused observation buffer, v2:66,223); `Class(num_envs=N)` is
the batched extension (torch tensors, obs float32[N,C,5,5] unless obs_mode="expanded").

Host side, by design: the single-env reset() placement runs the reference's rejection loops on
Python's `random` / `np.random` in the reference's own call order (v2:277-306; v2 draws goal and ball
BEFORE switching layout, v2:90-92; v4 after, v4:97-104), so seeded runs reproduce the reference's
placements.  Not reproduced: v1 opening ./visualize.txt at construction (v1:38), v4 printing the
visit map on every done (v4:269), cv2 / matplotlib output.
"""
import random

import numpy as np

from . import layouts as L
from ._staging import HostStaging
from .compat import Box, Discrete, Env
from .foveal_env import LmazeFovealVecEnv


def _f32_bits(x):
    return int(np.float32(x).view(np.uint32))


class _FovealBase(Env):
    metadata = {'render.modes': ['human']}
    _variant = "v2"

    def _setup(self, num_envs, device, obs_mode, seed):
        self.num_envs = int(num_envs)
        self._single = self.num_envs == 1
        self.obs_mode = obs_mode or ("expanded" if self._single else "compact")
        if self.obs_mode not in ("expanded", "compact"):
            raise ValueError("obs_mode must be 'expanded' or 'compact'")
        self._core = LmazeFovealVecEnv(self.num_envs, variant=self._variant, device=device,
                                       seed=0 if seed is None else int(seed), reset=False)
        self._stage = HostStaging(self._core.device) if self._single else None
        self._host = None

    @property
    def core(self):
        return self._core

    def _sync(self):
        if self._stage is not None:
            self._host = self._core.host_state(raw=self._stage.fetch(state=self._core._state)["state"].copy())
        else:
            self._host = self._core.host_state()
        return self._host

    def _step_single(self, action, local=False):
        """N = 1: upload the action, step, render x7, and bring the observation(s) + scalars back with one
        sync.  Returns ({"obs": view, "loc": view (v5/v6)}, host scalars); views are page-locked mirrors."""
        core, stage = self._core, self._stage
        expanded = self.obs_mode == "expanded"

        def body(act_dev):
            core.step(act_dev)
            want = {"state": core._state}
            if expanded:
                want["obs"] = core.expanded()[0]
                if local:
                    want["loc"] = core.expanded_local()[0]
            return want

        got = stage.step(("step", expanded, local), action, body)    # one hipGraph launch + one sync
        self._host = core.host_state(raw=got["state"].copy())
        return got, self._host

    def _reused(self, arr):
        """the reference's one shared output buffer (v2:66,123,223; v4:66,163,270)"""
        if getattr(self, "retStateExpanded", None) is None or self.retStateExpanded.shape != arr.shape:
            self.retStateExpanded = np.zeros(arr.shape, dtype=np.float32)
        np.copyto(self.retStateExpanded, arr)
        return self.retStateExpanded

    def _h(self, key):
        return (self._host if self._host is not None else self._sync())[key][0]

    def _reward_py(self, r32):
        table = getattr(self, "_rtab", None)
        if table is None:
            table = {_f32_bits(v): float(v) for v in (self.negativeNominal, self.positiveNominal, self.positiveFull)}
            table[_f32_bits(-0.0)] = -0.0
            table[_f32_bits(0.0)] = 0.0
            self._rtab = table
        return table.get(int(np.float32(r32).view(np.uint32)), float(r32))

    ball_x0 = property(lambda s: int(s._h("ball_xy")[0]))
    ball_y0 = property(lambda s: int(s._h("ball_xy")[1]))
    goal_x = property(lambda s: int(s._h("goal_xy")[0]))
    goal_y = property(lambda s: int(s._h("goal_xy")[1]))
    stepCount = property(lambda s: int(s._h("step_count")) if s._single else s._core.step_count)
    originalReward = property(lambda s: s._reward_py(s._h("reward")) if s._single else s._core.reward)

    def _obs_np(self, reuse):
        """single env, expanded mode: numpy (C,35,35); `reuse` = the reference's one shared buffer"""
        arr = self._stage.fetch(obs=self._core.expanded()[0])["obs"]
        return self._reused(arr) if reuse else arr.copy()

    def render(self, mode='human', close=False):
        if mode == 'human':
            self.VISUALIZE = True

    def rendering(self, msg):
        self.VISUALIZE = msg

    def writing(self, msg):
        self.SAVEFRAME = msg

    def setevaldir(self, msg):
        self.dirhead = msg


# ----------------------------------------------------------------------------------------
class LmazeEnv_v1(_FovealBase):
    """lmaze-v1: 14x14, 4 actions, 5x5 foveal window, two reward streams -- lmaze_env_v1.py:14-324."""
    _variant = "v1"

    def __init__(self, num_envs=1, device=None, obs_mode=None, seed=None):
        print("init-init")                                   # v1:20
        self.action_space = Discrete(4)                      # v1:21
        self.realgrid = 14
        self.expansionRatio = 7
        self.fovea = 5
        self.gridsize = self.fovea * self.expansionRatio
        self.observation_space = Box(0.0, 1.0, shape=(4, self.gridsize, self.gridsize))   # v1:26
        self.negativeNominal = -1.0                          # v1:27-29
        self.positiveNominal = 0.01
        self.positiveFull = 1.0
        self.RANDOM_BALL = False
        self.VISUALIZE = False
        self.grid = L.to_char_grid(L.V1_GRID_14)             # v1:40-53
        self._setup(num_envs, device, obs_mode, seed)
        self.reset()                                         # v1:57
        print("init-end")

    f_goal_x = property(lambda s: int(s._h("fgoal_xy")[0]))
    f_goal_y = property(lambda s: int(s._h("fgoal_xy")[1]))
    fovealStepCount = property(lambda s: int(s._h("foveal_step_count")) if s._single else s._core.foveal_step_count)
    fovealReward = property(lambda s: s._reward_py(s._h("foveal_reward")) if s._single else s._core.foveal_reward)

    def _out(self):
        if self.obs_mode == "compact":
            return self._core.obs
        return self._obs_np(reuse=False) if self._single else self._core.expanded()

    def reset(self, mask=None):
        self._core.reset(mask=mask)                          # ball = 'S' (v1:82-84), global view (v1:100)
        self._host = None
        return self._out()

    def setFovealGoal(self, msg0, msg1):                     # v1:104-110
        if self._single:
            ij = np.array([[int(msg0), int(msg1)]], np.int32)
        else:
            ij = np.stack([np.broadcast_to(np.asarray(msg0), (self.num_envs,)),
                           np.broadcast_to(np.asarray(msg1), (self.num_envs,))], axis=1).astype(np.int32)
        self._core.set_foveal_goal(ij)
        self._host = None
        return self._out()

    def step(self, msg):
        core = self._core
        if self._single:
            a = next((k for k in range(4) if msg == k), -1)  # v1:126-133 compares, never casts
            got, h = self._step_single(a)
            obs = got["obs"].copy() if "obs" in got else core.obs    # a fresh array per call (v1:258)
            return (obs, self._reward_py(h["reward"][0]), self._reward_py(h["foveal_reward"][0]),
                    bool(h["foveal_done"][0]), bool(h["done"][0]), msg)
        core.step(msg)
        self._host = None
        return self._out(), core.reward, core.foveal_reward, core.foveal_done, core.done, msg

    def getLocalView(self):                                  # v1:242-279 (what step()/setFovealGoal() return)
        return self._out()

    def getGlobalView(self):
        """v1:204-238: the 5x5 window around the ball of [ball, 'W', 'X', free] -- the global goal where getLocalView()
        shows the foveal goal --, x7; what reset() returns (v1:100).  Changes nothing but the returned buffer: rendered
        by the reset-mode launch of lmaze_foveal_reset (place = 0 keeps the ball) with the per-env scalars put back."""
        core = self._core
        snap = core._state.clone()
        core.reset(place=False)
        core._epoch -= 1                                     # no placement was drawn
        core._state.copy_(snap)
        self._host = None
        return self._out()

    @property
    def state(self):
        """v1:66-80 + the ball cell step() moves (v1:142-147): float32[4*G*G] -- ball one-hot, 'W', 'X', free = B|S|X --
        of the single env; float32[N, 4*G*G] (a torch tensor on the device) for a batch."""
        G = self.realgrid
        g = np.asarray(self.grid)
        static = np.stack([np.zeros((G, G)), g == 'W', g == 'X', np.isin(g, ['B', 'S', 'X'])]).astype(np.float32)
        if self._single:
            out = static.copy()
            out[0, self.ball_x0, self.ball_y0] = 1.0
            return out.reshape(-1)
        import torch
        core = self._core
        out = torch.from_numpy(static.reshape(-1)).to(core.device).repeat(self.num_envs, 1)
        idx = (core.ball_xy[:, 0] * G + core.ball_xy[:, 1]).long()
        out[torch.arange(self.num_envs, device=core.device), idx] = 1.0
        return out

    def initState(self):                                     # v1:289-290
        return self.state, self.originalReward, self.isEpisodeFinished(), {'newState': True}

    def isEpisodeFinished(self, queryType="plain"):          # v1:294-304
        if self._single:
            return bool(self.originalReward == self.positiveFull or self.stepCount == 200)
        return (self._core.reward == self.positiveFull) | (self._core.step_count == 200)

    def isFovealEpisodeFinished(self):                       # v1:308-324
        if self._single:
            return bool(self._h("foveal_done"))
        return self._core.foveal_done


# ----------------------------------------------------------------------------------------
class _TeleportBase(_FovealBase):
    """shared body of v2 / v4 (25-way action, five layouts)."""

    def _common_init(self, channel, num_envs, device, obs_mode, seed):
        self.expansionRatio = 7
        self.fovea = 5
        self.gridsize = self.fovea * self.expansionRatio
        self.channel = channel
        self.actionChannel = 1
        self.stateChannel = 2 * channel + 1
        self.step_limit = 50
        self.fovealStepCount = 0
        self.negativeNominal = -1.0
        self.positiveNominal = -0.01
        self.positiveFull = 100.0
        self.RANDOM_BALL = True
        self.RANDOM_GOAL = True
        self.AUTO_VISUALIZE = False
        self.SAVEFRAME = False
        self.VISUALIZE = False
        self.localDone = False
        self.dir = "."
        self.dirhead = "eval_"
        self.retStateExpanded = None
        self._tables = [L.to_char_grid(t) for t in L.FOVEAL_GRIDS_18]
        self._lid = 0
        self._setup(num_envs, device, obs_mode, seed)

    @property
    def grid(self):
        return self._tables[self._lid]

    @property
    def realgrid(self):
        return self.grid.shape[0]

    f_goal_x = property(lambda s: s._fgoal[0])
    f_goal_y = property(lambda s: s._fgoal[1])

    # the reference's public placement helpers (v2:277-306), same draw order
    def setGoal(self):
        g = self.grid
        x, y = 0, 0
        if self.RANDOM_GOAL:
            while g[x][y] == 'W' or g[x][y] == 'S':
                x = random.randint(1, self.realgrid - 2)
                y = random.randint(1, self.realgrid - 2)
        else:
            s = np.where(g == 'X')
            x, y = int(s[0][0]), int(s[1][0])
        self._goal = (x, y)

    def setBall(self):
        g = self.grid
        x, y = 0, 0
        if self.RANDOM_BALL:
            while g[x][y] == 'W' or g[x][y] == 'X' or (self._goal[0] == x and self._goal[1] == y):
                x = random.randint(1, self.realgrid - 2)
                y = random.randint(1, self.realgrid - 2)
        else:
            s = np.where(g == 'S')
            x, y = int(s[0][0]), int(s[1][0])
        self._ball = (x, y)

    def setGrid(self):
        self._lid = int(np.random.randint(1, 6)) - 1         # v2:306

    def _upload_and_reset(self):
        self._core.set_state(ball_xy=np.array([self._ball], np.int32), goal_xy=np.array([self._goal], np.int32),
                             layout_id=np.array([self._lid], np.int32))
        self._core.reset(place=False)
        self._host = None
        self._fgoal = (0, 0)

    def _out(self):
        if self.obs_mode == "compact":
            return self._core.obs
        return self._obs_np(reuse=True) if self._single else self._core.expanded()

    def step(self, goal):
        core = self._core
        if self._single:
            g = int(goal)                                    # v2:131
            if not 0 <= g < 25:
                raise IndexError("index %d is out of bounds for the 5x5 action plane (lmaze_env_v2.py:135-136)" % g)
            bx, by = self.ball_x0, self.ball_y0
            self._fgoal = (bx + g // 5 - 2, by + g % 5 - 2)  # v2:151-152
            got, h = self._step_single(g)
            obs = self._reused(got["obs"]) if "obs" in got else core.obs
            return obs, self._reward_py(h["reward"][0]), bool(h["done"][0]), g
        core.step(goal)
        self._host = None
        return self._out(), core.reward, core.done, goal


class LmazeEnv_v2(_TeleportBase):
    """lmaze-v2: five 18x18 layouts, Discrete(25), obs (5,35,35) -- lmaze_env_v2.py:17-435."""
    _variant = "v2"

    def __init__(self, num_envs=1, device=None, obs_mode=None, seed=None):
        self.state_type = "twoState"                         # v2:24
        self._common_init(2, num_envs, device, obs_mode, seed)
        self.observation_space = Box(0.0, 1.0, shape=(self.stateChannel, self.gridsize, self.gridsize))   # v2:38
        self.action_space = Discrete(self.fovea * self.fovea)                                            # v2:39
        if self._single:
            self.setGrid()                                   # v2:72
        self.reset()                                         # v2:75

    def reset(self, mask=None):
        if self._single:
            self.setGoal()                                   # v2:90-92: goal and ball on the CURRENT grid,
            self.setBall()                                   #           then the layout switches
            self.setGrid()
            self._upload_and_reset()
        else:
            self._core.reset(mask=mask)
            self._host = None
        return self._out()


class LmazeEnv_v4(_TeleportBase):
    """lmaze-v4: v2 + visit-map plane, obs (7,35,35); declares no spaces upstream (v4:42-44)."""
    _variant = "v4"

    def __init__(self, num_envs=1, device=None, obs_mode=None, seed=None):
        self.state_type = "twoState-threeLayers"             # v4:23
        self._common_init(3, num_envs, device, obs_mode, seed)
        self.reset()                                         # v4:69

    def reset(self, mask=None):
        if self._single:
            self.setGrid()                                   # v4:97-104
            self.setGoal()
            self.setBall()
            self._upload_and_reset()
        else:
            self._core.reset(mask=mask)
            self._host = None
        return self._out()

    @property
    def state(self):
        """float32 [3,G,G] of env 0: free, goal one-hot, visit map (v4:112-119)."""
        g = self.grid
        out = np.zeros((3, self.realgrid, self.realgrid), np.float32)
        out[0] = np.isin(g, ['B', 'S', 'X'])
        out[1, self.goal_x, self.goal_y] = 1.0
        out[2] = self._core.visit[0].cpu().numpy()
        return out


# ----------------------------------------------------------------------------------------
class LmazeEnv_v5(_TeleportBase):
    """lmaze-v5: two-level planner / local env, plannerStep() + step(), 8-tuple return --
    lmaze_env_v5.py:17-712.  Declares no spaces upstream (v5:41-43)."""
    _variant = "v5"

    def __init__(self, num_envs=1, device=None, obs_mode=None, seed=None):
        self.state_type = "twoState-threeLayers"             # v5:23
        self._common_init(3, num_envs, device, obs_mode, seed)
        self.step_limit = 10                                 # v5:45-46
        self.foveal_step_limit = 50
        self.reset()                                         # v5:80

    # reference attribute names (v5:62-78)
    f_goal_x0 = property(lambda s: int(s._h("fgoal_xy")[0]))
    f_goal_y0 = property(lambda s: int(s._h("fgoal_xy")[1]))
    ball_x1 = property(lambda s: int(s._h("ball1_xy")[0]))
    ball_y1 = property(lambda s: int(s._h("ball1_xy")[1]))
    fovea_x0 = property(lambda s: int(s._h("fovea_xy")[0]))
    fovea_y0 = property(lambda s: int(s._h("fovea_xy")[1]))
    fovea_x1 = property(lambda s: int(s._h("fovea_xy")[2]))
    fovea_y1 = property(lambda s: int(s._h("fovea_xy")[3]))
    fovealStepCount = property(lambda s: int(s._h("foveal_step_count")) if s._single else s._core.foveal_step_count,
                               lambda s, v: None)
    globalReward = property(lambda s: s._reward_py(s._h("reward")) if s._single else s._core.reward)
    originalReward = property(lambda s: s._reward_py(s._h("foveal_reward")) if s._single else s._core.foveal_reward)
    globalDone = property(lambda s: bool(s._h("done")) if s._single else s._core.done)
    localDone = property(lambda s: bool(s._h("foveal_done")) if s._single else s._core.foveal_done, lambda s, v: None)

    @property
    def fovealGoal(self):
        """float32 (1,5,5) one-hot plane (v5:166-169)."""
        if self._single:
            p = np.zeros((1, 5, 5), np.float32)
            p.reshape(-1)[int(self._h("foveal_goal"))] = 1.0
            return p
        return self._core.obs_local[:, 3:4]

    def _fov(self):
        if self.obs_mode == "compact":
            return self._core.obs
        if self._single:
            return self._stage.fetch(obs=self._core.expanded()[0])["obs"].copy()              # fresh array, v5:308
        return self._core.expanded()

    def _loc(self):
        if self.obs_mode == "compact":
            return self._core.obs_local
        if self._single:
            return self._stage.fetch(loc=self._core.expanded_local()[0])["loc"].copy()
        return self._core.expanded_local()

    def reset(self, mask=None):
        if self._single:
            self.setGrid()                                   # v5:105, then goal, ball (v5:114-115)
            self.setGoal()
            self.setBall()
            self._upload_and_reset()
        else:
            self._core.reset(mask=mask)
            self._host = None
        return self._fov()

    def plannerStep(self, goal):                             # v5:158-182
        if self._single:
            g = int(goal)
            if not 0 <= g < 25:
                raise IndexError("index %d is out of bounds for the 5x5 fovealGoal plane (lmaze_env_v5.py:169)" % g)
            self._core.planner_step(np.array([g], np.int32))
        else:
            self._core.planner_step(goal)
        self._host = None
        return self._loc()

    def step(self, goal):                                    # v5:187-292
        core = self._core
        if self._single:
            a = int(goal)
            got, h = self._step_single(a if -2 ** 31 <= a < 2 ** 31 else -1, local=True)
            # the reference's buildLocalObservation indexes a 5x5 frame with ball - fovea_1 + 2 (v5:364-365)
            for b in (h["ball_xy"][0], h["ball1_xy"][0]):
                for k in (0, 1):
                    idx = int(b[k]) - int(h["fovea_xy"][0][2 + k]) + 2
                    if not -5 <= idx < 5:
                        raise IndexError("index %d is out of bounds for axis with size 5 (lmaze_env_v5.py:364-365)" % idx)
            fov = got["obs"].copy() if "obs" in got else core.obs            # fresh arrays, v5:308,359
            loc = got["loc"].copy() if "loc" in got else core.obs_local
            return (fov, loc, self._reward_py(h["reward"][0]), self._reward_py(h["foveal_reward"][0]),
                    bool(h["done"][0]), bool(h["foveal_done"][0]), self.fovealGoal, a)
        core.step(goal)
        self._host = None
        return (self._fov(), self._loc(), core.reward, core.foveal_reward, core.done, core.foveal_done,
                self.fovealGoal, goal)

    def hierStep(self, planner_goal, action):
        """Batched extension (num_envs > 1; not in the reference): what the usual v5/v6 loop does per env -- reset() when
        globalDone, plannerStep(planner_goal) when localDone or after that reset, step(action) -- as ONE launch
        (lmaze_v5_hier_step).  Returns step()'s 8-tuple."""
        if self._single:
            raise ValueError("hierStep is the batched extension: construct with num_envs > 1 (a single env uses reset / plannerStep / step)")
        core = self._core
        core.hier_step(action, planner_goal)
        self._host = None
        return (self._fov(), self._loc(), core.reward, core.foveal_reward, core.done, core.foveal_done,
                self.fovealGoal, action)

    def buildFovealObservation(self):
        return self._fov()

    def buildLocalObservation(self):
        return self._loc()

    def safeFovealGoal(self):                                # v5:501-502: a stub upstream
        print("return a safe foveal goal for training local agent ")


class LmazeEnv_v6(LmazeEnv_v5):
    """lmaze-v6: v5 + safeFovealGoal() (lmaze_env_v6.py:505-523)."""
    _variant = "v6"

    def safeFovealGoal(self):
        if not self._single:
            return self._core.safe_foveal_goal()
        g = self.grid
        bx, by = self.ball_x0, self.ball_y0
        small = g[bx - 2:bx + 3, by - 2:by + 3]              # v6:510, same np.random draws as the reference
        a = np.random.randint(0, 25)
        while small[int(a / 5)][a % 5] == 'W':
            a = np.random.randint(0, 25)
        return a
