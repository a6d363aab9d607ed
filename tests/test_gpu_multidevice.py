"""GPU, more than one process: shards on separate ranks reproduce the single-device batch, and the library's
reductions run over the process group.  `nccl` (= RCCL) with one MI355X per rank needs >= 2 devices and is
skipped on a one-GPU lease; the same worker over gloo with the ranks sharing the device runs everywhere."""
import importlib
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = importlib.import_module("gym-lmaze_amd")
L = PKG.layouts


def _run_ranks(tmp_path, world, backend, total=100003, T=130):
    sys.path.insert(0, ROOT)
    import bench
    rc = bench.spawn_ranks([sys.executable, os.path.join(ROOT, "tests", "_rank_worker.py"), str(tmp_path), str(total),
                            str(T), backend], world)
    assert rc == 0
    # the whole batch on this process's device 0
    env = PKG.LmazeVecEnv(total, variant="v0", layout=L.to_codes(L.V0_GRID_12), device="cuda:0", seed=17)
    acts = np.random.RandomState(3).randint(0, 4, (T, total)).astype(np.int32)
    env.rollout(torch.from_numpy(acts).to("cuda:0"), auto_reset=True)
    h, obs = env.host_state(), env.obs.cpu().numpy()
    seen, sums = 0, dict(done=0, goal_rewards=0, done_steps=0, goal_count=0)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "shard%d.npz" % r))
        s, c = int(d["start"]), d["ball_xy"].shape[0]
        assert s == seen
        for k in ("ball_xy", "step_count", "goal_count", "done"):
            assert (d[k] == h[k][s:s + c]).all(), (k, r)
        assert (d["reward"].view(np.uint32) == h["reward"][s:s + c].view(np.uint32)).all(), r
        assert (d["obs"] == obs[s:s + c]).all(), r
        seen += c
        st = json.load(open(os.path.join(str(tmp_path), "stats%d.json" % r)))
        assert st["world"] == world and st["backend"] == backend
        for k in sums:
            sums[k] += st["local"][k]
    assert seen == total
    whole = env.episode_stats()
    for r in range(world):
        st = json.load(open(os.path.join(str(tmp_path), "stats%d.json" % r)))
        assert st["all"] == sums == whole, (st["all"], sums, whole)       # the all_reduce on every rank
    assert whole["goal_count"] > 0 and int(h["step_count"].max()) <= 100   # episodes ended and restarted
    return [json.load(open(os.path.join(str(tmp_path), "stats%d.json" % r)))["device"] for r in range(world)]


def test_two_ranks_sharing_the_device_over_gloo(tmp_path):
    devices = _run_ranks(tmp_path, 2, "gloo")
    assert len(devices) == 2


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X: one process per GPU over RCCL")
@pytest.mark.parametrize("world", [2, 4, 8])
def test_one_rank_per_gpu_over_rccl(tmp_path, world):
    if torch.cuda.device_count() < world:
        pytest.skip("%d devices" % torch.cuda.device_count())
    devices = _run_ranks(tmp_path, world, "nccl")
    assert devices == list(range(world))         # one device per rank
