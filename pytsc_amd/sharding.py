"""Replica sharding across the GPUs of one node.

Replicas are fully independent, so the step path has NO collective: each rank
(one process per GPU) owns a contiguous block of replicas and its own engine
handle.  torch.distributed is used only off the step path: the barrier that
brackets a timed region and the max-over-ranks of the elapsed time.
"""
import os

import torch
import torch.distributed as dist


def world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(
        os.environ.get("WORLD_SIZE", "1"))


def shard_replicas(total, world_size, rank):
    """Contiguous block [start, start+count) of `total` replicas for `rank`."""
    base, rem = divmod(total, world_size)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def init(backend=None):
    rank, local_rank, world_size = world()
    if world_size > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world_size, **kw)
    return rank, local_rank, world_size


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
