"""CPU, world_size 2 over gloo: the multi-process harness bench.py uses (rendezvous on
127.0.0.1, barrier, MAX-over-ranks reduction, per-rank shard + env_base), with the C oracle
standing in for the device so the shard bookkeeping is exercised without a GPU: two ranks'
shards, gathered, equal one process holding the whole batch."""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_rollout(O, lay, start, count, seed, acts):
    G = lay.shape[0]
    p = O.params(O.VARIANT_V0, G, O.LAYOUT_SHARED)
    ball = np.zeros((count, 2), np.int32)
    sc = np.zeros(count, np.int32)
    rew = np.zeros(count, np.float32)
    done = np.zeros(count, np.uint8)
    gc = np.zeros(count, np.int32)
    obs = np.zeros((count, G, G), np.int32)
    O.reset(p, lay, None, seed, 0, ball, None, sc, rew, done, obs, env_base=start)
    for t in range(acts.shape[0]):
        O.step_v0(p, lay, np.ascontiguousarray(acts[t, start:start + count]), ball, sc, rew, done, gc, obs)
    return ball, rew, obs


def _worker(rank, world, port, total, T, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    S = importlib.import_module("gym-lmaze_amd.sharding")
    L = importlib.import_module("gym-lmaze_amd.layouts")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, count = S.shard_range(total, rank, world)
    lay = L.to_codes(L.V0_GRID_12)
    acts = np.random.RandomState(3).randint(0, 4, (T, total)).astype(np.int32)   # same on every rank
    dist.barrier()
    ball, rew, obs = _oracle_rollout(O, lay, start, count, 17, acts)
    # the bench's reduction: elapsed time -> MAX over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    # gather shard sizes and a checksum of each shard's planes on rank 0
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([count], dtype=torch.int64))
    assert sum(int(s.item()) for s in sizes) == total
    np.savez(os.path.join(out_dir, "shard%d.npz" % rank), ball=ball, rew=rew, obs=obs, start=start)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    O.build()
    L = importlib.import_module("gym-lmaze_amd.layouts")
    total, T, world = 1001, 12, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, T, str(tmp_path)), nprocs=world, join=True)
    lay = L.to_codes(L.V0_GRID_12)
    acts = np.random.RandomState(3).randint(0, 4, (T, total)).astype(np.int32)
    ball, rew, obs = _oracle_rollout(O, lay, 0, total, 17, acts)
    seen = 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "shard%d.npz" % r))
        s, c = int(d["start"]), d["ball"].shape[0]
        assert (d["ball"] == ball[s:s + c]).all() and (d["obs"] == obs[s:s + c]).all()
        assert (d["rew"].view(np.uint32) == rew[s:s + c].view(np.uint32)).all()
        seen += c
    assert seen == total


def _reduce_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    S = importlib.import_module("gym-lmaze_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert S.max_over_ranks(10.0 + rank) == 10.0 + world - 1
    tot = S.sum_over_ranks([rank + 1, 5, 100 * rank])
    assert tot.tolist() == [sum(r + 1 for r in range(world)), 5 * world, sum(100 * r for r in range(world))]
    dist.destroy_process_group()


def test_bench_reductions_world2():
    """the two reductions the multi-GPU bench / episode statistics use, over gloo with 2 ranks"""
    S = importlib.import_module("gym-lmaze_amd.sharding")
    assert S.max_over_ranks(3.5) == 3.5 and S.sum_over_ranks([1, 2]).tolist() == [1, 2]   # no process group
    mp.spawn(_reduce_worker, args=(2, _free_port()), nprocs=2, join=True)
