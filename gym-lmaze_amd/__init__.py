"""gym-lmaze_amd: MI355X-native batched L-maze step path (drop-in for gkm2708/gym-lmaze).

The directory name carries a hyphen, so import it with
    importlib.import_module("gym-lmaze_amd")
or, as a user of the reference would, through the alias package `gym_lmaze`
(`import gym_lmaze; gym_lmaze.make("lmaze-v0")`, `from gym_lmaze.envs import LmazeEnv`).

Importing this package loads liblmaze_hip.so and raises if it is missing: the HIP kernels
are the only implementation of the step path.
"""
from . import _abi  # noqa: F401  (fails loudly when the HIP library is absent)
from . import layouts  # noqa: F401
from .compat import make, register, registered_ids  # noqa: F401
from .envs import LmazeEnv, LmazeEnv_v3  # noqa: F401
from .foveal_env import FOVEAL_VARIANTS, LmazeFovealVecEnv  # noqa: F401
from .foveal_envs import LmazeEnv_v1, LmazeEnv_v2, LmazeEnv_v4, LmazeEnv_v5, LmazeEnv_v6  # noqa: F401
from .sharding import gather_over_ranks, max_over_ranks, shard_range, sum_over_ranks  # noqa: F401
from .vec_env import VARIANTS, LmazeVecEnv  # noqa: F401

__all__ = ["LmazeVecEnv", "LmazeFovealVecEnv", "FOVEAL_VARIANTS", "LmazeEnv", "LmazeEnv_v1", "LmazeEnv_v2", "LmazeEnv_v3", "LmazeEnv_v4", "LmazeEnv_v5", "LmazeEnv_v6", "VARIANTS", "layouts", "make", "register",
           "registered_ids", "shard_range"]
