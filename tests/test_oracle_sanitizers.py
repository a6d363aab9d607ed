"""CPU: the oracle's C code under AddressSanitizer + UBSan: the fixture replays run
in a child process against oracle/liblmaze_oracle_asan.so with libasan preloaded; any out-of-bounds access,
misaligned access or signed overflow aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_replays_clean_under_asan_ubsan():
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "liblmaze_oracle_asan.so"])
    env = dict(os.environ)
    env.update({"LD_PRELOAD": libasan, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1",
                "UBSAN_OPTIONS": "halt_on_error=1", "LMAZE_ORACLE_LIB": os.path.join(ROOT, "oracle", "liblmaze_oracle_asan.so")})
    sel = "v0_g8 or v0_g32 or v3_g11 or v1_scripted or v2_seed1 or v4_noreset or v5_seed1 or v6_seed2 or philox"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q",
                          "-p", "no:cacheprovider", "-k", sel], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "passed" in out.stdout and "ERROR: AddressSanitizer" not in out.stdout and "runtime error" not in out.stdout
    # the two-level composition (malloc'ed masks) and the statistics rollout as well
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_properties.py"), "-x", "-q",
                          "-p", "no:cacheprovider", "-k", "two_level"], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "passed" in out.stdout and "ERROR: AddressSanitizer" not in out.stdout and "runtime error" not in out.stdout
