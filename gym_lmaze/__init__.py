"""Alias package: the import name and env ids of the reference (gym_lmaze/__init__.py:1-38),
resolved to the MI355X-native implementation in ../gym-lmaze_amd/.

    import gym_lmaze                      # registers lmaze-v0 ... with gym/gymnasium when present
    env = gym_lmaze.make("lmaze-v0")      # built-in registry, works without gym

`lmaze-v7` is registered upstream but its module does not exist there
(gym_lmaze/envs/__init__.py:8 imports a missing file), so it is not registered here.
"""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.isdir(os.path.join(_root, "gym-lmaze_amd")):
    # a checkout: the implementation sits next to this package under its hyphenated directory name
    if _root not in sys.path:
        sys.path.insert(0, _root)
    _impl = importlib.import_module("gym-lmaze_amd")
else:
    # `pip install .` (setup.py): the same package under the import name gym_lmaze_amd
    _impl = importlib.import_module("gym_lmaze_amd")

make = _impl.make
registered_ids = _impl.registered_ids

_ENTRY_POINTS = {
    "lmaze-v0": "gym_lmaze.envs:LmazeEnv",
    "lmaze-v1": "gym_lmaze.envs:LmazeEnv_v1",
    "lmaze-v2": "gym_lmaze.envs:LmazeEnv_v2",
    "lmaze-v3": "gym_lmaze.envs:LmazeEnv_v3",
    "lmaze-v4": "gym_lmaze.envs:LmazeEnv_v4",
    "lmaze-v5": "gym_lmaze.envs:LmazeEnv_v5",
    "lmaze-v6": "gym_lmaze.envs:LmazeEnv_v6",
}
for _id, _ep in _ENTRY_POINTS.items():
    _impl.register(_id, _ep)
