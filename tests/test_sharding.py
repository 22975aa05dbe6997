"""CPU tests of the N>1 path: channel sharding (i mod G) and the
max-over-ranks timing of the benchmark contract, with two gloo ranks."""
import os
import socket

import pytest


def test_shard_partition_properties():
    from gnsscorr import sharding
    for world in (1, 2, 3, 4, 8):
        owned = [sharding.shard_channels(256, world, r) for r in range(world)]
        flat = sorted(c for o in owned for c in o)
        assert flat == list(range(256))
        assert all(sharding.owner_of(c, world) == r for r, o in enumerate(owned) for c in o)
        assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1
    assert sharding.shard_channels(256, 8, 3)[:3] == [3, 11, 19] and len(sharding.shard_channels(256, 8, 3)) == 32
    assert sharding.weak_shard(32, 8, 7)[:2] == [7, 15]
    with pytest.raises(ValueError):
        sharding.shard_channels(8, 2, 2)
    assert sharding.max_over_ranks(1.5) == 1.5
    assert sharding.aggregate_throughput(100.0, 10, 8, 2.0) == 4000.0
    assert sharding.gather_per_rank({"rank": 0, "x": 1.0}) == [{"rank": 0, "x": 1.0}]


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "gnss-sdr-1_amd"))
    import torch.distributed as dist
    from gnsscorr import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.weak_shard(4, world, rank)
    dist.barrier()
    elapsed = 0.010 * (rank + 1)  # rank 1 is the slow one
    worst = sharding.max_over_ranks(elapsed, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    # the per-GPU figures of the bench line: every rank's own record, in rank order, on every rank
    per_gpu = sharding.gather_per_rank({"rank": rank, "elapsed_s": elapsed, "kernel_ms": 0.25 + 0.01 * rank}, dist, world)
    q.put((rank, mine, worst, gathered, per_gpu))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_max_time():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, m0, w0, g0, p0), (r1, m1, w1, g1, p1) = res
    assert m0 == [0, 2, 4, 6] and m1 == [1, 3, 5, 7]
    assert w0 == w1 == pytest.approx(0.020)
    assert sorted(c for part in g0 for c in part) == list(range(8))
    assert p0 == p1 and [g["rank"] for g in p0] == [0, 1]
    assert p0[1]["kernel_ms"] == pytest.approx(0.26) and max(p0, key=lambda g: g["elapsed_s"])["rank"] == 1
