"""TEST INFRASTRUCTURE ONLY (oracle/): loads the upstream reference's env classes.

Build-container only.  /root/reference does not exist on the GPU box, so nothing in
tests/, bench.py or __graft_entry__ imports this module at run time; its single use is
`oracle/gen_golden.py`, which writes the committed fixtures under tests/golden/.

Why a loader is needed: every reference env file starts with
`import gym` / `import cv2` (gym_lmaze/envs/lmaze_env.py:1-6) and neither is installed
here, and the reference package __init__ imports a module that does not exist
(gym_lmaze/envs/__init__.py:8).  We therefore put *inert* stand-ins for `gym` and `cv2`
into sys.modules and load each env file directly by path.  The stand-ins carry no
arithmetic: `gym.Env` is an empty base class, `spaces.Box/Discrete` only remember
their constructor arguments, and `cv2` is reached only when VISUALIZE/SAVEFRAME are
switched on (off by default, lmaze_env.py:26).  The transition and observation code
that runs is the reference's own, unmodified, read from /root/reference.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("LMAZE_REFERENCE_ROOT", "/root/reference")
_ENV_DIR = os.path.join(REF_ROOT, "gym_lmaze", "envs")

_FILES = {
    "v0": ("lmaze_env.py", "LmazeEnv"),
    "v1": ("lmaze_env_v1.py", "LmazeEnv_v1"),
    "v2": ("lmaze_env_v2.py", "LmazeEnv_v2"),
    "v3": ("lmaze_env_v3.py", "LmazeEnv_v3"),
    "v4": ("lmaze_env_v4.py", "LmazeEnv_v4"),
    "v5": ("lmaze_env_v5.py", "LmazeEnv_v5"),
    "v6": ("lmaze_env_v6.py", "LmazeEnv_v6"),
}


def available():
    return os.path.isdir(_ENV_DIR)


class _Space:
    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs
        self.shape = kwargs.get("shape")
        self.n = args[0] if args and isinstance(args[0], int) else None


def _install_stubs():
    if "gym" in sys.modules and getattr(sys.modules["gym"], "_lmaze_stub", False):
        return
    gym = types.ModuleType("gym")
    gym._lmaze_stub = True

    class Env(object):
        metadata = {}

    gym.Env = Env
    spaces = types.ModuleType("gym.spaces")
    spaces.Box = type("Box", (_Space,), {})
    spaces.Discrete = type("Discrete", (_Space,), {})
    gym.spaces = spaces
    gym.error = types.ModuleType("gym.error")
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    utils.seeding = seeding
    gym.utils = utils
    sys.modules.update({
        "gym": gym, "gym.spaces": spaces, "gym.error": gym.error,
        "gym.utils": utils, "gym.utils.seeding": seeding,
    })
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    os.environ.setdefault("MPLBACKEND", "Agg")


_cache = {}


def load_class(variant):
    """Return the reference env class for 'v0'..'v6' (its own code, loaded by path)."""
    if variant in _cache:
        return _cache[variant]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    _install_stubs()
    fname, cname = _FILES[variant]
    # dont_write_bytecode: the reference tree is read-only for us
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        spec = importlib.util.spec_from_file_location("_lmaze_ref_" + variant,
                                                      os.path.join(_ENV_DIR, fname))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = old
    _cache[variant] = getattr(mod, cname)
    return _cache[variant]


def make(variant, quiet=True):
    """Construct a reference env.  v1 opens ./visualize.txt at init (lmaze_env_v1.py:38),
    so we run its constructor from a scratch directory that holds an empty one."""
    import contextlib
    import io
    import tempfile
    cls = load_class(variant)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "visualize.txt"), "w").close()
        os.chdir(tmp)
        try:
            if quiet:
                with contextlib.redirect_stdout(io.StringIO()):
                    env = cls()
            else:
                env = cls()
        finally:
            os.chdir(cwd)
    return env
