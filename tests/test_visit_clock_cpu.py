"""CPU: the clock-relative visit-map arithmetic the kernels use (gym-lmaze_amd/csrc/lmaze_visit.h, compiled here for the
host) against the reference's eager whole-plane recurrence (lmaze_env_v4.py:211-214), bit for bit, incl. the subnormal
range and renormalisation.  The GPU tests check the same through lmaze_foveal_materialise_visit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("vc") / "libvisit_clock_host.so")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-Wall", "-Wextra", "-fPIC", "-shared", "-o", so,
                           os.path.join(HERE, "csrc", "visit_clock_host.c")])
    lib = C.CDLL(so)
    lib.visit_clock_fuzz.restype = C.c_int64
    lib.visit_clock_fuzz.argtypes = [C.c_uint64, C.c_int64] + [C.POINTER(C.c_int64)] * 3
    lib.vc_true.restype = C.c_uint32
    lib.vc_true.argtypes = [C.c_uint32, C.c_int]
    lib.vc_add.restype = C.c_uint32
    lib.vc_add.argtypes = [C.c_uint32, C.c_int]
    return lib


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_lazy_clock_equals_the_eager_recurrence(host, seed):
    compared, sub, ren = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    bad = host.visit_clock_fuzz(seed, 3_000_000, C.byref(compared), C.byref(sub), C.byref(ren))
    assert bad == 0
    assert compared.value == 3_000_000 and sub.value > 1000 and ren.value > 100     # the hard regimes were reached


def test_known_answers(host):
    f = lambda x: int(np.float32(x).view(np.uint32))
    # reset (zeros, clock 0) + one window: 0.5 under clock 1 is stored as 2^-126
    s = host.vc_add(0, 0)
    assert s == 0x00800000 and host.vc_true(s, 1) == f(0.5)
    assert host.vc_true(s, 2) == f(0.25) and host.vc_true(s, 126) == f(2.0 ** -126)
    assert host.vc_true(s, 127) == f(2.0 ** -127)            # subnormal: still exact (one bit)
    assert host.vc_true(s, 149) == 1 and host.vc_true(s, 150) == 0 and host.vc_true(s, 400) == 0   # 2^-150 ties to even: 0
    # three visits in a row (0.875), then decay: 7 * 2^-152 -> RNE on every step, not once
    s3 = host.vc_add(host.vc_add(s, 1), 2)
    assert host.vc_true(s3, 3) == f(0.875)
    v = np.float32(0.875)
    for E in range(4, 160):
        v = np.float32(np.float64(v) / 2.0)
        assert host.vc_true(s3, E) == int(v.view(np.uint32)), E
    # saturation at 1.0
    s1, E = 0, 0
    for _ in range(40):
        s1, E = host.vc_add(s1, E), E + 1
    assert host.vc_true(s1, E) == f(1.0)
