"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

The hot path shards by samples: every rank holds a full replica, runs forward/backward on its own
images and the only exchange is ONE sum-all-reduce of the flat fp32 gradient buffer per step
(98 MB for PHiSeg 7/5), scaled by 1/world_size - identical to the full-batch mean-loss gradient
for equal shard sizes (BatchNorm statistics stay per replica, as in DDP).  SURVEY.md 8e.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, local_rank, world_size); a single process needs no group."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    elif world > 1 and torch.cuda.is_available() and dist.get_backend() == "nccl":
        torch.cuda.set_device(local_rank)
    return rank, local_rank, world


def mean_scalar(t, group=None):
    """Mean of a 0-d tensor over the ranks (identity for a single process)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    v = t.detach().clone().reshape(1).to(torch.float32)
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    return (v / dist.get_world_size(group)).reshape(())


def shard_bounds(n, rank, world):
    """Contiguous [lo, hi) slice of n samples owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_mean_(flat, group=None):
    """In-place average of a flat gradient buffer over the ranks of `group`."""
    if not dist.is_initialized():
        return flat
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / world)
    return flat


def broadcast_(flat, src=0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat
