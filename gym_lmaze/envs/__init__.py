"""Entry-point module of the registered ids (reference: gym_lmaze/envs/__init__.py)."""
from .. import _impl
LmazeEnv = _impl.LmazeEnv
LmazeEnv_v3 = _impl.LmazeEnv_v3
LmazeEnv_v1 = _impl.LmazeEnv_v1
LmazeEnv_v2 = _impl.LmazeEnv_v2
LmazeEnv_v4 = _impl.LmazeEnv_v4
LmazeEnv_v5 = _impl.LmazeEnv_v5
LmazeEnv_v6 = _impl.LmazeEnv_v6
