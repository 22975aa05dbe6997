"""Channel sharding over the GPUs of one node.

Channels are independent given the IQ block: channel i runs on GPU i mod G,
acquisition jobs likewise.  There is no data-path collective; RCCL (or gloo in
the CPU tests) is only used for the start barrier and for the max-over-ranks of
the elapsed time that the benchmark contract asks for.
"""


def shard_channels(n_total, world, rank):
    """Strong scaling view: the global channel ids owned by `rank` (i mod G)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return list(range(rank, n_total, world))


def weak_shard(per_gpu, world, rank):
    """Weak scaling view (fixed channels per GPU): global ids rank, rank+G, ..."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return [rank + world * j for j in range(per_gpu)]


def owner_of(channel, world):
    return channel % world


def max_over_ranks(value, dist=None, device=None):
    """MAX all-reduce of a python float (identity without a process group)."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_per_rank(record, dist=None, world=1):
    """Every rank's small record (a dict of python scalars), in rank order, on every
    rank: the per-GPU figures of the multi-GPU bench line.  Identity without a group."""
    if dist is None or not dist.is_initialized():
        return [record]
    out = [None] * world
    dist.all_gather_object(out, record)
    return sorted(out, key=lambda r: r["rank"])


def aggregate_throughput(units_per_rank_per_step, steps, world, elapsed_max_s):
    """Whole-job throughput: units all ranks processed / max-over-ranks time."""
    return units_per_rank_per_step * world * steps / elapsed_max_s
