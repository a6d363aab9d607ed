"""CPU, build container only (skipped where /root/reference is absent, e.g. on the GPU box): the committed fixtures
are what the committed generator emits -- byte for byte, whichever generators run before -- so a recording can
always be traced to oracle/gen_golden.py running the reference's own reset()/step()."""
import filecmp
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/gym_lmaze/envs"),
                                reason="the reference tree only exists in the build container")


def _generate(out_dir, *which):
    env = dict(os.environ, LMAZE_GOLDEN_OUT=str(out_dir))
    subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), *which], check=True, env=env,
                   capture_output=True, timeout=1500)
    return sorted(f for f in os.listdir(str(out_dir)) if f.endswith(".npz"))


def test_v5_and_v6_recordings_do_not_depend_on_generator_order(tmp_path):
    """Round 1's v5_seed0 only came out as committed when `gen_golden.py v5` ran alone (the layout cache consumed
    draws from the freshly seeded global streams).  Alone, and after another generator has filled that cache: the
    same bytes as the committed files."""
    alone, after = tmp_path / "alone", tmp_path / "after"
    alone.mkdir(); after.mkdir()
    a = _generate(alone, "v5", "v6")
    b = _generate(after, "v4", "v6", "v5")
    assert set(a) <= set(b) and "v5_seed0.npz" in a and "v6_seed0.npz" in a
    for f in b:
        assert filecmp.cmp(os.path.join(str(after), f), os.path.join(GOLDEN, f), shallow=False), f
    for f in a:
        assert filecmp.cmp(os.path.join(str(alone), f), os.path.join(GOLDEN, f), shallow=False), f


def test_every_committed_fixture_has_a_generator():
    text = open(os.path.join(ROOT, "oracle", "gen_golden.py")).read()
    for f in os.listdir(GOLDEN):
        if f.endswith(".npz"):
            assert '"%s"' % f[:-4] in text or f == "reset_hist.npz", f
