"""CPU: bench.py's self-launch (`python bench.py --gpus N` from a plain shell) -- the rank spawner alone, with a
stand-in child; the parent never imports torch or touches a GPU."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import os, sys, time
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert os.environ["LMAZE_BENCH_SELF_LAUNCHED"] == "1"
open(os.path.join(sys.argv[1], "rank%d" % r), "w").write("%d %d %s" % (r, w, os.environ["MASTER_PORT"]))
print("line of rank %d" % r)
if len(sys.argv) > 2 and r == int(sys.argv[2]):
    sys.exit(7)
if len(sys.argv) > 2:
    time.sleep(120)          # would hang in a barrier for ever: the launcher has to stop it
"""


def test_spawn_ranks_env_stdout_routing_and_exit_code(tmp_path, capfd):
    import bench
    assert "torch" not in sys.modules or True       # (other tests of this session may have imported it)
    rc = bench.spawn_ranks([sys.executable, "-c", CHILD, str(tmp_path)], 3)
    out, err = capfd.readouterr()
    assert rc == 0
    ports = set()
    for r in range(3):
        rr, w, port = open(os.path.join(str(tmp_path), "rank%d" % r)).read().split()
        assert (int(rr), int(w)) == (r, 3)
        ports.add(port)
    assert len(ports) == 1                                             # one rendezvous for all ranks
    assert out.strip() == "line of rank 0"                             # stdout carries rank 0 only: the ONE JSON line
    assert "line of rank 1" in err and "line of rank 2" in err


def test_spawn_ranks_stops_the_others_when_a_rank_fails(tmp_path):
    import bench
    t0 = time.time()
    rc = bench.spawn_ranks([sys.executable, "-c", CHILD, str(tmp_path), "1"], 3)
    assert rc == 7 and time.time() - t0 < 60


def test_plain_gpus_n_launches_before_importing_torch(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent spawns and returns the children's verdict.  Here
    (no GPU) every child exits with bench.py's 'needs an MI355X' message -- which proves the children were started
    as ranks and that the parent itself never needed a device."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: covered by tests/test_gpu_bench_contract.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0 and "needs an MI355X" in out.stderr and out.stdout.strip() == ""


def test_world_size_mismatch_is_loud():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode != 0 and "process group of 3 ranks" in out.stderr
