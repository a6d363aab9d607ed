"""gym / gymnasium are optional: the reference needs `gym` only for a base class, two space
declarations and the id registry (gym_lmaze/__init__.py:1-38, lmaze_env.py:11-20).  When
neither is importable (as in the build image) a minimal stand-in with the same attribute
names is used, plus a built-in registry exposing make(id).
"""
import importlib

import numpy as np

_gym = None
for _name in ("gym", "gymnasium"):
    try:
        _gym = importlib.import_module(_name)
        break
    except Exception:  # not installed
        _gym = None

if _gym is not None:
    Env = _gym.Env
    Box = _gym.spaces.Box
    Discrete = _gym.spaces.Discrete
    GYM_NAME = _gym.__name__
else:
    GYM_NAME = None

    class Env(object):
        metadata = {}
        action_space = None
        observation_space = None

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())

        def __repr__(self):
            return "Box(%s, %s, %s, %s)" % (self.low, self.high, self.shape, self.dtype)

    class Discrete(object):
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def sample(self):
            return int(np.random.randint(self.n))

        def contains(self, x):
            try:
                return 0 <= int(x) < self.n
            except (TypeError, ValueError):
                return False

        def __repr__(self):
            return "Discrete(%d)" % self.n


# ---- registry: ids -> entry points, the drop-in contract of gym_lmaze/__init__.py:3-38 ----
_REGISTRY = {}


def register(id, entry_point):
    _REGISTRY[id] = entry_point
    if _gym is not None:
        try:
            _gym.envs.registration.register(id=id, entry_point=entry_point)
        except Exception:  # already registered / registry API drift: the built-in registry still works
            pass


def registered_ids():
    return sorted(_REGISTRY)


def make(id, **kwargs):
    """gym.make() stand-in: resolve 'module:Class' and construct it."""
    if id not in _REGISTRY:
        raise KeyError("no registered env with id %r (have %s)" % (id, registered_ids()))
    mod, cls = _REGISTRY[id].split(":")
    return getattr(importlib.import_module(mod), cls)(**kwargs)
