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


def synthetic_volume(in_ch, num_classes, dhw, seed=20201005):
    """BraTS-like synthetic volume (BASELINE configs[4]): (1, in_ch, D, H, W) image channels N(0, 0.25^2) clipped to +-0.5, a
    label volume of nested balls (labels 0 .. num_classes-1) returned one-hot (1, K, D, H, W) for forward() - the form the
    reference's BraTS path hands over, utils.py:296-298 - and as a label map (1, 1, D, H, W) for loss()."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d, h, w = dhw
    x = np.clip(rng.standard_normal((1, in_ch, d, h, w)).astype(np.float32) * 0.25, -0.5, 0.5)
    centre = [rng.uniform(0.35, 0.65) * n for n in dhw]
    zz, yy, xx = np.ogrid[0:d, 0:h, 0:w]
    dist2 = (zz - centre[0]) ** 2 + (yy - centre[1]) ** 2 + (xx - centre[2]) ** 2
    labels = np.zeros((d, h, w), np.float32)
    for k in range(1, num_classes):
        labels[dist2 <= (min(dhw) * 0.45 / k) ** 2] = k
    onehot = np.stack([(labels == k) for k in range(num_classes)]).astype(np.float32)[None]
    return x, onehot, labels[None, None]
