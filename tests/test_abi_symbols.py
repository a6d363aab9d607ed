"""CPU: liblmaze_hip.so loads and exports exactly what include/lmaze.h declares; argument
checks answer before any launch (no compute without a GPU)."""
import ctypes as C
import importlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lmaze.h")


@pytest.fixture(scope="module")
def abi():
    lib = os.path.join(ROOT, "gym-lmaze_amd", "liblmaze_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "gym-lmaze_amd", "csrc"), "-s"])
    return importlib.import_module("gym-lmaze_amd._abi")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lmaze_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(abi):
    names = declared_functions()
    assert names == sorted(abi.SYMBOLS)
    for n in names:
        assert hasattr(abi.lib, n), n


def test_header_cites_reference_lines():
    text = open(HEADER).read()
    for needle in ("v0:146-237", "v3:220-402", "v0:217-234", "v0:246-249"):
        assert needle in text


def test_abi_version_and_errors(abi):
    assert abi.lib.lmaze_abi_version() == abi.ABI_VERSION == 4
    assert abi.strerror(0) == "ok"
    p = abi.make_params(abi.VARIANT_V0, 12, abi.LAYOUT_SHARED, 100, -1.0, -0.01, 100.0)
    # NULL pointers / bad sizes are rejected before anything is launched
    assert abi.lib.lmaze_step_v0(None, None, None, None, None, None, None, None, None, 1, None) == -1
    assert abi.lib.lmaze_step_v0(C.byref(p), None, None, None, None, None, None, None, None, 1, None) == -1
    bad = abi.make_params(abi.VARIANT_V0, 2, abi.LAYOUT_SHARED, 100, -1.0, -0.01, 100.0)
    assert abi.lib.lmaze_step_v0(C.byref(bad), None, None, None, None, None, None, None, None, 1, None) == -2
    assert abi.lib.lmaze_step_v3(C.byref(p), 16, 16, 16, 16, 16, 16, 16, None, 1, None) == -3
    assert abi.lib.lmaze_step_v0(C.byref(p), 16, 16, 16, 16, 16, 16, None, None, -1, None) == -5
    assert abi.lib.lmaze_step_v0(C.byref(p), 16, 16, 4, 16, 16, 16, None, None, 1, None) == -6
    m = (C.c_int32 * 4)(1, 2, 4, 8)
    assert abi.lib.lmaze_render_expanded(16, 12, 99, m, 4, 16, 1, None) == -7
    assert "aligned" in abi.strerror(-6)
    # a count no launch could cover is refused, never narrowed into a truncated grid (include/lmaze.h LMAZE_MAX_ENVS)
    assert abi.lib.lmaze_step_v0(C.byref(p), 16, 16, 16, 16, 16, 16, None, None, (1 << 30) + 1, None) == -5
    assert abi.lib.lmaze_observe(C.byref(p), 16, 16, None, 16, 1 << 40, None) == -5


def test_params_struct_layout(abi):
    assert C.sizeof(abi.LmazeParams) == 32
    assert [f[0] for f in abi.LmazeParams._fields_] == ["variant", "grid", "layout_mode", "step_limit",
                                                        "reward_wall", "reward_move", "reward_goal", "launch_hint"]


def test_no_device_is_loud(abi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pkg = importlib.import_module("gym-lmaze_amd")
    with pytest.raises(RuntimeError):
        pkg.LmazeVecEnv(4)
    with pytest.raises(RuntimeError):
        pkg.LmazeVecEnv(4, device="cpu")


def test_product_never_touches_oracle():
    """The package may not import, load or name anything under oracle/."""
    pkg = os.path.join(ROOT, "gym-lmaze_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(base, f)).read()
                assert "oracle" not in src.lower(), os.path.join(base, f)


def test_header_is_plain_c(tmp_path):
    """include/lmaze.h must compile as C99 (the boundary is a C ABI, not a C++ one), and a C caller that
    fills both parameter structs must agree with the ctypes mirrors on their sizes."""
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include "lmaze.h"\n'
                   'int main(void){ LmazeParams p = {0}; LmazeFovealParams f = {0}; LmazeFovealBuffers b = {0};\n'
                   '  (void)p; (void)f; (void)b;\n'
                   '  printf("%zu %zu %zu\\n", sizeof p, sizeof f, sizeof b); return 0; }\n')
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)], text=True).split()]
    abi_mod = importlib.import_module("gym-lmaze_amd._abi")
    assert sizes == [C.sizeof(abi_mod.LmazeParams), C.sizeof(abi_mod.LmazeFovealParams), C.sizeof(abi_mod.LmazeFovealBuffers)]
