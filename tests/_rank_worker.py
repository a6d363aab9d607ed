"""One rank of the multi-device tests (started by bench.spawn_ranks; not a test module).

    python tests/_rank_worker.py <out_dir> <total_envs> <steps> <backend>

Builds this rank's shard of a `total_envs` batch (LmazeVecEnv(count, env_base=start)), rolls it out with the
fused auto-reset on the shard's columns of one seeded action tensor, and leaves its state in <out_dir>/shard<r>.npz;
then checks the two reductions the library issues over the process group: episode_stats(all_ranks=True) (a SUM of
four int64) and max_over_ranks / gather_over_ranks.  backend "nccl" (= RCCL, one device per rank) or "gloo"
(ranks share the visible device: the rehearsal that runs on a one-GPU box)."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, total, T, backend = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    local = int(os.environ["LOCAL_RANK"]) % ndev if backend != "nccl" else int(os.environ["LOCAL_RANK"])
    assert local < ndev, "rank %d has no device" % rank
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)                               # RCCL's banner goes to stderr
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    dist.barrier()
    os.dup2(saved, 1)
    os.close(saved)
    pkg = importlib.import_module("gym-lmaze_amd")
    start, count = pkg.shard_range(total, rank, world)
    lay = pkg.layouts.to_codes(pkg.layouts.V0_GRID_12)
    env = pkg.LmazeVecEnv(count, variant="v0", layout=lay, device=dev, seed=17, env_base=start)
    acts = np.random.RandomState(3).randint(0, 4, (T, total)).astype(np.int32)          # same on every rank
    a = torch.from_numpy(np.ascontiguousarray(acts[:, start:start + count])).to(dev)
    env.rollout(a, auto_reset=True)
    torch.cuda.synchronize()
    h = env.host_state()
    np.savez(os.path.join(out_dir, "shard%d.npz" % rank), start=start, obs=env.obs.cpu().numpy(),
             **{k: np.array(v) for k, v in h.items()})
    local_stats = env.episode_stats()
    all_stats = env.episode_stats(all_ranks=True)
    t_max = pkg.max_over_ranks(10.0 + rank, device=dev if backend == "nccl" else None)
    t_all = pkg.gather_over_ranks(10.0 + rank, device=dev if backend == "nccl" else None)
    assert t_max == 10.0 + world - 1 and t_all == [10.0 + r for r in range(world)], (t_max, t_all)
    json.dump({"local": local_stats, "all": all_stats, "device": local, "backend": dist.get_backend(),
               "world": dist.get_world_size()}, open(os.path.join(out_dir, "stats%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("ranks ok")


if __name__ == "__main__":
    main()
