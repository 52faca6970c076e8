"""Data-parallel path on CPU: world_size-2 gloo process group, flat-gradient averaging, sharding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import unet_zoo_amd  # noqa: F401
from unet_zoo_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    # DP semantics: the averaged buffer equals the mean of the per-shard gradients (SURVEY 8e parity gate)
    dp.allreduce_mean_(flat)
    params = torch.full((8,), float(rank))
    dp.broadcast_(params, src=0)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), torch.cat([flat, params]).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_mean_world2_gloo(tmp_path):
    world, n = 2, 1000
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    ref = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    for a in got:
        assert np.allclose(a[:n], ref.numpy(), atol=1e-6)
        assert np.all(a[n:] == 0.0)                 # parameters broadcast from rank 0
    assert np.array_equal(got[0], got[1])           # replicas stay bitwise identical


def test_shard_bounds_cover_batch():
    for n, world in [(32, 8), (32, 3), (5, 8), (1, 2)]:
        spans = [dp.shard_bounds(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_single_process_is_a_noop():
    t = torch.arange(4.0)
    assert dp.allreduce_mean_(t.clone()).equal(t)
