"""CPU: shipped layout tables equal the layout bytes read off the reference objects (stored
in the golden fixtures), and host-side validation / registry / sharding logic."""
import importlib

import numpy as np
import pytest

from helpers import load_golden

L = importlib.import_module("gym-lmaze_amd.layouts")


def test_tables_match_reference_layouts():
    assert (L.to_codes(L.V0_GRID_12) == load_golden("v0_c1_g12_seed0")["layout"]).all()
    assert (L.to_codes(L.V3_GRID_18) == load_golden("v3_g18_seed0")["layout"]).all()
    assert (L.to_codes(L.GRID_8_BORDERED) == load_golden("v0_g8_seed0")["layout"]).all()
    assert (L.to_codes(L.open_room(11, (5, 5))) == load_golden("v0_g11_open_seed0")["layout"]).all()
    assert (L.to_codes(L.V1_GRID_14) == load_golden("v1_seed0")["layout"]).all()
    five = load_golden("v2_seed0")["layouts"]
    assert (five == load_golden("v4_seed0")["layouts"]).all()
    for k, t in enumerate(L.FOVEAL_GRIDS_18):
        assert (L.to_codes(t) == five[k]).all(), k


def test_all_tables_valid():
    for t in (L.V0_GRID_12, L.V1_GRID_14, L.V3_GRID_18, L.GRID_8_BORDERED) + L.FOVEAL_GRIDS_18:
        c = L.to_codes(t)
        assert c.shape[0] == c.shape[1]
        L.validate(c)
        assert (c == ord("S")).sum() == 1 and (c == ord("X")).sum() == 1


def test_validate_rejects_open_border_and_bad_cells():
    g = L.to_codes(L.V0_GRID_12).copy()
    g[0, 3] = ord("B")
    with pytest.raises(ValueError):
        L.validate(g)
    g = L.to_codes(L.V0_GRID_12).copy()
    g[3, 3] = ord("Q")
    with pytest.raises(ValueError):
        L.validate(g)
    g = L.to_codes(L.V0_GRID_12).copy()
    g[g == ord("X")] = ord("B")
    with pytest.raises(ValueError):
        L.validate(g)
    L.validate(g, need_goal_marker=False)
    with pytest.raises(ValueError):
        L.to_codes(np.zeros((3, 4), np.uint8))


def test_registry_ids_and_entry_points():
    import gym_lmaze
    assert gym_lmaze.registered_ids() == ["lmaze-v%d" % k for k in range(7)]
    from gym_lmaze.envs import LmazeEnv, LmazeEnv_v3
    assert LmazeEnv.__name__ == "LmazeEnv" and LmazeEnv_v3.__name__ == "LmazeEnv_v3"
    with pytest.raises(KeyError):
        gym_lmaze.make("lmaze-v7")   # unresolvable upstream too (envs/__init__.py:8)


def test_shard_range_partitions():
    S = importlib.import_module("gym-lmaze_amd.sharding")
    for total in (0, 1, 7, 8, 1000003, 1 << 23):
        for world in (1, 2, 3, 8):
            spans = [S.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0
            assert sum(c for _, c in spans) == total
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        S.shard_range(10, 2, 2)


def test_setup_py_installs_alias_package_implementation_and_library(tmp_path):
    """The reference installs with `pip install -e .` (its setup.py:1-6).  Ours: setup.py builds liblmaze_hip.so and
    installs `gym_lmaze` + the implementation (import name gym_lmaze_amd) with the .so as package data; the
    installed copy imports from anywhere and registers the ids."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prefix = tmp_path / "p"
    subprocess.run([sys.executable, "setup.py", "-q", "build", "--build-base", str(tmp_path / "b"), "install", "--prefix",
                    str(prefix), "--single-version-externally-managed", "--record", str(tmp_path / "rec.txt")],
                   cwd=root, check=True, capture_output=True, timeout=900)
    site = [os.path.join(b, "site-packages") for b, d, _ in os.walk(str(prefix)) if "site-packages" in d][0]
    assert os.path.exists(os.path.join(site, "gym_lmaze_amd", "liblmaze_hip.so"))
    code = ("import gym_lmaze, gym_lmaze.envs as E; assert gym_lmaze._impl.__name__ == 'gym_lmaze_amd'; "
            "assert gym_lmaze.registered_ids() == ['lmaze-v%d' % k for k in range(7)]; "
            "assert E.LmazeEnv.__module__.startswith('gym_lmaze_amd'); print('ok')")
    out = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=dict(os.environ, PYTHONPATH=site),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]
    for junk in ("build", "gym_lmaze.egg-info"):
        import shutil
        shutil.rmtree(os.path.join(root, junk), ignore_errors=True)
