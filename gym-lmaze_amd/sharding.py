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
