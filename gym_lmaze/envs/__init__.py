"""Entry-point module of the registered ids (reference: gym_lmaze/envs/__init__.py)."""
import importlib

_impl = importlib.import_module("gym-lmaze_amd")
LmazeEnv = _impl.LmazeEnv
LmazeEnv_v3 = _impl.LmazeEnv_v3
