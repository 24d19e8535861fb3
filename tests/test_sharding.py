"""N > 1 path on CPU: replica sharding + the barrier / max-over-ranks that bracket
a timed region, world_size 2 over gloo.  No collective exists on the step path."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from pytsc_amd import sharding

    r, lr, w = sharding.init(backend="gloo")
    start, count = sharding.shard_replicas(65536 + 3, w, r)
    sharding.barrier()
    t = sharding.max_over_ranks(1.0 + r)
    total = sharding.sum_over_ranks(count * 16 * 10)
    q.put((r, start, count, t, total))
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, c0, t0, tot0), (r1, s1, c1, t1, tot1) = res
    assert (s0, c0) == (0, 32770) and (s1, c1) == (32770, 32769)  # contiguous, covers all replicas
    assert t0 == t1 == 2.0  # max over ranks
    assert tot0 == tot1 == (65536 + 3) * 16 * 10  # whole-job units


def test_shards_partition_the_replicas():
    from pytsc_amd.sharding import shard_replicas

    for total in (1, 7, 4096, 65536):
        for w in (1, 2, 4, 8):
            spans = [shard_replicas(total, w, r) for r in range(w)]
            assert sum(c for _, c in spans) == total
            pos = 0
            for s, c in spans:
                assert s == pos
                pos += c
