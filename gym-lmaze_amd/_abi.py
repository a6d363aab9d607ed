"""ctypes binding of liblmaze_hip.so -- the C ABI declared in include/lmaze.h.

This is the only way the package reaches the kernels, and there is no other compute path:
if the library is missing or does not load, importing this module raises.  Build it with
`python -c "import __graft_entry__ as g; g.build()"` or `make -C gym-lmaze_amd/csrc`.
"""
import ctypes as C
import os

# torch FIRST: its wheel bundles its own libamdhip64 / libhsa-runtime64, and liblmaze_hip.so needs the same
# sonames.  Loaded in this order the dynamic linker gives both ONE HIP runtime (torch's); loaded the other
# way round the process ends up with two HSA runtimes and the second one finds no device (every launch then
# fails with hipErrorNoDevice) -- seen on MI355X with `import gym_lmaze` before `import torch`.
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LMAZE_HIP_LIB") or os.path.join(HERE, "liblmaze_hip.so")   # override: another build of the same ABI

ABI_VERSION = 4
VARIANT_V0, VARIANT_V3 = 0, 3
VARIANT_V1, VARIANT_V2, VARIANT_V4, VARIANT_V5, VARIANT_V6 = 1, 2, 4, 5, 6
FOVEA = 5
LAYOUT_SHARED, LAYOUT_PER_ENV = 0, 1
OBS_BALL, OBS_WALL, OBS_GOAL, OBS_FREE = 1, 2, 4, 8
MAX_GRID, MAX_CHANNELS = 64, 8

# every symbol include/lmaze.h declares (tests/test_abi_symbols.py parses the header and
# checks this list and the loaded library against it)
SYMBOLS = ("lmaze_abi_version", "lmaze_strerror", "lmaze_device_info", "lmaze_step_v0", "lmaze_step_v3",
           "lmaze_step_v0_autoreset", "lmaze_step_v3_autoreset", "lmaze_observe", "lmaze_reset",
           "lmaze_episode_stats", "lmaze_bandwidth_probe", "lmaze_render_expanded", "lmaze_foveal_step", "lmaze_foveal_step_autoreset", "lmaze_foveal_reset", "lmaze_v1_set_foveal_goal",
           "lmaze_v5_planner_step", "lmaze_v5_hier_step", "lmaze_v6_safe_foveal_goal", "lmaze_expand_planes",
           "lmaze_foveal_visit_bytes", "lmaze_foveal_materialise_visit", "lmaze_foveal_load_visit",
           "lmaze_describe_step", "lmaze_describe_foveal_step", "lmaze_rollout",
           "lmaze_step_u8", "lmaze_observe_u8")


class LmazeParams(C.Structure):
    """struct LmazeParams of include/lmaze.h (constants the reference hard-codes in __init__)."""
    _fields_ = [("variant", C.c_int32), ("grid", C.c_int32), ("layout_mode", C.c_int32),
                ("step_limit", C.c_int32), ("reward_wall", C.c_float), ("reward_move", C.c_float),
                ("reward_goal", C.c_float), ("launch_hint", C.c_int32)]


class LmazeFovealParams(C.Structure):
    """struct LmazeFovealParams of include/lmaze.h."""
    _fields_ = [("variant", C.c_int32), ("grid", C.c_int32), ("n_layouts", C.c_int32), ("step_limit", C.c_int32),
                ("foveal_step_limit", C.c_int32), ("reward_wall", C.c_float), ("reward_move", C.c_float),
                ("reward_goal", C.c_float), ("launch_hint", C.c_int32)]


FOVEAL_BUFFER_FIELDS = ("ball_xy", "goal_xy", "fgoal_xy", "layout_id", "step_count", "foveal_step_count",
                        "reward", "foveal_reward", "done", "foveal_done", "visit", "obs",
                        "ball1_xy", "fovea_xy", "last_xy", "foveal_goal", "obs_local", "visit_clock")


class LmazeFovealBuffers(C.Structure):
    """struct LmazeFovealBuffers of include/lmaze.h: device pointers, one element per env."""
    _fields_ = [(n, C.c_void_p) for n in FOVEAL_BUFFER_FIELDS]


class LmazeError(RuntimeError):
    def __init__(self, fn, code):
        self.code = code
        RuntimeError.__init__(self, "%s failed: %d (%s)" % (fn, code, strerror(code)))


def _sources():
    """csrc/*.hip, csrc/*.h and include/lmaze.h: what liblmaze_hip.so is built from."""
    src = os.path.join(HERE, "csrc")
    out = [os.path.join(src, f) for f in sorted(os.listdir(src)) if f.endswith((".hip", ".h"))] if os.path.isdir(src) else []
    hdr = os.path.join(os.path.dirname(HERE), "include", "lmaze.h")
    return out + ([hdr] if os.path.exists(hdr) else [])


def _load():
    if not os.path.exists(LIB_PATH):
        # never built behind the caller's back: under torchrun every rank of a fresh checkout would race to
        # compile and link the same files, and a silent build hides a toolchain problem
        raise ImportError(
            "gym-lmaze_amd: %s is missing. The HIP library is the only compute path (no CPU fallback); "
            "build it with `make -C %s` (hipcc, --offload-arch=gfx950) or `python -c 'import __graft_entry__ as g; "
            "g.build()'`." % (LIB_PATH, os.path.join(HERE, "csrc")))
    if "LMAZE_HIP_LIB" not in os.environ:
        stale = [f for f in _sources() if os.path.getmtime(f) > os.path.getmtime(LIB_PATH)]
        if stale:
            import warnings
            warnings.warn("gym-lmaze_amd: %s is older than %s -- rebuild with `make -C %s`"
                          % (os.path.basename(LIB_PATH), ", ".join(os.path.basename(f) for f in stale),
                             os.path.join(HERE, "csrc")), RuntimeWarning, stacklevel=3)
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    P = C.POINTER(LmazeParams)
    lib.lmaze_abi_version.restype = C.c_int
    lib.lmaze_abi_version.argtypes = []
    lib.lmaze_strerror.restype = C.c_char_p
    lib.lmaze_strerror.argtypes = [C.c_int]
    lib.lmaze_device_info.restype = C.c_int
    lib.lmaze_device_info.argtypes = [C.c_int, C.POINTER(i32), C.c_char_p, i32]
    lib.lmaze_step_v0.restype = C.c_int
    lib.lmaze_step_v0.argtypes = [P, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    lib.lmaze_step_v3.restype = C.c_int
    lib.lmaze_step_v3.argtypes = [P, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    lib.lmaze_step_v0_autoreset.restype = C.c_int
    lib.lmaze_step_v0_autoreset.argtypes = [P, vp, vp, vp, vp, vp, vp, vp, vp, i64, u64, u64, i64, vp, vp, vp]
    lib.lmaze_step_v3_autoreset.restype = C.c_int
    lib.lmaze_step_v3_autoreset.argtypes = [P, vp, vp, vp, vp, vp, vp, vp, vp, i64, u64, u64, i64, vp, vp, vp]
    lib.lmaze_observe.restype = C.c_int
    lib.lmaze_observe.argtypes = [P, vp, vp, vp, vp, i64, vp]
    lib.lmaze_reset.restype = C.c_int
    lib.lmaze_reset.argtypes = [P, vp, vp, u64, u64, i64, vp, vp, vp, vp, vp, vp, i64, vp]
    lib.lmaze_render_expanded.restype = C.c_int
    lib.lmaze_render_expanded.argtypes = [vp, i32, i32, C.POINTER(i32), i32, vp, i64, vp]
    lib.lmaze_bandwidth_probe.restype = C.c_int
    lib.lmaze_bandwidth_probe.argtypes = [vp, vp, i64, vp]
    lib.lmaze_episode_stats.restype = C.c_int
    lib.lmaze_episode_stats.argtypes = [vp, vp, vp, vp, C.c_float, i64, vp, vp]
    FP, FB = C.POINTER(LmazeFovealParams), C.POINTER(LmazeFovealBuffers)
    lib.lmaze_foveal_step.restype = C.c_int
    lib.lmaze_foveal_step.argtypes = [FP, vp, vp, FB, i64, vp]
    lib.lmaze_foveal_step_autoreset.restype = C.c_int
    lib.lmaze_foveal_step_autoreset.argtypes = [FP, vp, vp, FB, i64, u64, u64, i64, vp, vp, vp]
    lib.lmaze_foveal_reset.restype = C.c_int
    lib.lmaze_foveal_reset.argtypes = [FP, vp, vp, i32, u64, u64, i64, FB, i64, vp]
    lib.lmaze_v1_set_foveal_goal.restype = C.c_int
    lib.lmaze_v1_set_foveal_goal.argtypes = [FP, vp, vp, vp, FB, i64, vp]
    lib.lmaze_v5_planner_step.restype = C.c_int
    lib.lmaze_v5_planner_step.argtypes = [FP, vp, vp, vp, FB, i64, vp]
    lib.lmaze_v5_hier_step.restype = C.c_int
    lib.lmaze_v5_hier_step.argtypes = [FP, vp, vp, vp, FB, i64, u64, u64, i64, vp, vp, vp]
    lib.lmaze_v6_safe_foveal_goal.restype = C.c_int
    lib.lmaze_v6_safe_foveal_goal.argtypes = [FP, vp, u64, u64, i64, FB, vp, i64, vp]
    lib.lmaze_expand_planes.restype = C.c_int
    lib.lmaze_expand_planes.argtypes = [vp, i32, i32, i32, vp, i64, vp]
    lib.lmaze_foveal_visit_bytes.restype = i64
    lib.lmaze_foveal_visit_bytes.argtypes = [i32, i64]
    lib.lmaze_foveal_materialise_visit.restype = C.c_int
    lib.lmaze_foveal_materialise_visit.argtypes = [FP, FB, vp, i64, vp]
    lib.lmaze_foveal_load_visit.restype = C.c_int
    lib.lmaze_foveal_load_visit.argtypes = [FP, FB, vp, i64, vp]
    lib.lmaze_step_u8.restype = C.c_int
    lib.lmaze_step_u8.argtypes = [P, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, u64, u64, i64, vp, vp, vp]
    lib.lmaze_observe_u8.restype = C.c_int
    lib.lmaze_observe_u8.argtypes = [P, vp, vp, vp, vp, vp, i64, vp]
    lib.lmaze_rollout.restype = C.c_int
    lib.lmaze_rollout.argtypes = [P, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, u64, u64, i64, vp]
    lib.lmaze_describe_step.restype = C.c_int
    lib.lmaze_describe_step.argtypes = [P, i64, i32, i32, C.c_char_p, i32]
    lib.lmaze_describe_foveal_step.restype = C.c_int
    lib.lmaze_describe_foveal_step.argtypes = [FP, i64, i32, C.c_char_p, i32]
    if lib.lmaze_abi_version() != ABI_VERSION:
        raise ImportError("liblmaze_hip.so ABI %d != binding %d: rebuild" % (lib.lmaze_abi_version(), ABI_VERSION))
    return lib


lib = _load()


def strerror(code):
    return lib.lmaze_strerror(int(code)).decode("ascii", "replace")


def check(fn, code):
    if code != 0:
        raise LmazeError(fn, code)


def device_info(device=0):
    cu = C.c_int32(0)
    name = C.create_string_buffer(64)
    check("lmaze_device_info", lib.lmaze_device_info(int(device), C.byref(cu), name, 64))
    return {"cu_count": cu.value, "arch": name.value.decode("ascii", "replace")}


def describe_step(params, n, auto_reset=False, with_obs=True):
    """The kernel / grid / launch policy the library would queue for n envs with these LmazeParams (lmaze_describe_step)."""
    buf = C.create_string_buffer(256)
    check("lmaze_describe_step", lib.lmaze_describe_step(C.byref(params), int(n), 1 if auto_reset else 0,
                                                         2 if with_obs == "u8" else (1 if with_obs else 0), buf, 256))
    return buf.value.decode("ascii", "replace")


def describe_foveal_step(params, n, auto_reset=False):
    buf = C.create_string_buffer(256)
    check("lmaze_describe_foveal_step", lib.lmaze_describe_foveal_step(C.byref(params), int(n), 1 if auto_reset else 0, buf, 256))
    return buf.value.decode("ascii", "replace")


def make_params(variant, grid, layout_mode, step_limit, reward_wall, reward_move, reward_goal):
    return LmazeParams(int(variant), int(grid), int(layout_mode), int(step_limit), float(reward_wall),
                       float(reward_move), float(reward_goal), 0)
