"""Synthetic LIDC-like batches for benchmarking and smoke runs: images N(0, 0.25^2)
clipped to [-0.5, 0.5] (the range lidc_data_loader.py:92 produces), binary random-disc masks, and
optional standard-normal latent noise.  numpy PCG64 with a fixed seed, so every box sees the same
bytes (the parity tests assert it agrees with the oracle's generator)."""
import numpy as np


def synthetic_batch(batch, height=128, width=128, seed=20201004, eps_shapes=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    x = np.clip(rng.standard_normal((batch, 1, height, width)).astype(np.float32) * 0.25, -0.5, 0.5)
    yy, xx = np.mgrid[0:height, 0:width]
    mask = np.zeros((batch, 1, height, width), np.float32)
    for b in range(batch):
        r = rng.uniform(8, 24) * min(height, width) / 128.0
        cy, cx = rng.uniform(r, height - r), rng.uniform(r, width - r)
        mask[b, 0] = ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r).astype(np.float32)
    eps = None
    if eps_shapes is not None:
        eps = [rng.standard_normal(s).astype(np.float32) for s in eps_shapes]
    return x, mask, eps
