"""Multi-GPU sharding of the env batch: contiguous blocks of envs per rank, no collective
on the step path (envs are independent).  One process per GPU;
each rank builds `LmazeVecEnv(count, env_base=start, ...)` for its own block, so reset
draws are keyed by the GLOBAL env index and a sharded run reproduces a single-device run
of the whole batch bit for bit."""


def shard_range(num_envs_total, rank, world_size):
    """[start, start+count) of the global env ids owned by `rank` (remainder spread over the
    first ranks, so counts differ by at most one)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, rem = divmod(int(num_envs_total), int(world_size))
    start = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    return start, count


def max_over_ranks(value, device=None):
    """MAX of a per-rank scalar over the process group (the elapsed time of a timed region): the only
    reduction the multi-GPU bench needs.  No process group (single process) -> the value itself.
    Works on any backend: nccl (= RCCL) with a device tensor, gloo on the CPU."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device=None):
    """SUM of a small vector of per-rank counters (episode statistics): off the step
    path, a handful of scalars per call."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(values, dtype=torch.int64, device=device if device is not None else "cpu").clone()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def gather_over_ranks(value, device=None):
    """The per-rank scalar of every rank, in rank order (the bench reports min / max of the per-rank step time
    so that a straggler shows).  No process group -> [value]."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
