"""Mini-batch provider - drop-in for the reference's ``data/batch_provider.py`` ``BatchProvider`` (same constructor keywords,
``next_batch`` :43-67, ``iterate_batches`` :69-99, sampling without replacement over the whole index set, random annotator per
sample :131-137, per-sample random augmentation :140-271), re-designed for a 288 GB GPU:

  * the split's arrays are uploaded ONCE and stay resident in HBM (LIDC: 1.2 GB);
  * a batch costs the host only the random draws - indices, annotators and the augmentation parameters, taken from numpy's
    global RNG in exactly the reference's order, so a seeded run consumes the same random stream - and one small parameter upload;
  * gather + annotator selection + rotation + crop/resize + flips are ONE HIP kernel (csrc/augment.hip) that writes the
    (B, 1, H, W) image and (B, H, W) label tensors the model consumes: no per-image Python loop, no cv2, no host copy of pixels.

``next_batch`` returns device tensors (``UNetModel.train`` passes them straight to ``forward``); ``next_batch(bs, host=True)``
returns numpy arrays like the reference.  Without a GPU the provider still samples (index / parameter logic is host code) but
cannot assemble pixels - there is no CPU fallback for the product path.

Quirks kept: the reference reads the flip switches as ``do_fliplr`` / ``do_flipud`` while its experiment files set
``do_flip_lr`` / ``do_flip_ud`` (phiseg_7_5_12.py:32-36), so flips never fire from those files - same here; ``normalise_images``
is called but its result discarded (:117-118), i.e. a no-op.  Not built: ``do_elasticaug`` (no experiment file enables it) and
``resize_to`` (UZH only) raise NotImplementedError.
"""
import ctypes as C

import numpy as np
import torch

from .. import _ffi


def draw_augmentation(n_images, shape, options):
    """The random parameters of ``_augmentation_function`` (:186-266) for `n_images` images of `shape` = (n_x, n_y), drawn from
    numpy's global RNG in the reference's order.  Returns a float32 (n, 8) table:
    {do_rot, cos, sin, do_scale, p_x, p_y, r, flips (bit 0 = lr, bit 1 = ud)}."""
    def opt(name, default):
        return options[name] if name in options else default
    do_rot, do_scale = opt("do_rotations", False), opt("do_scaleaug", False)
    do_lr, do_ud = opt("do_fliplr", False), opt("do_flipud", False)
    if opt("do_elasticaug", False):
        raise NotImplementedError("elastic augmentation is not part of the native input pipeline")
    nth = opt("augment_every_nth", 2)
    if (do_rot or do_scale) and not opt("nlabels", None):
        raise AssertionError("When doing augmentations with rotations, scaling, or elastic transformations "
                             "the parameter 'nlabels' must be provided.")
    n_x, n_y = shape
    out = np.zeros((n_images, 8), np.float32)
    out[:, 1] = 1.0
    for ii in range(n_images):
        if np.random.randint(nth) == 0:
            if do_rot:
                ang = np.random.uniform(-opt("rot_degrees", 10.0), opt("rot_degrees", 10.0))
                out[ii, 0], out[ii, 1], out[ii, 2] = 1.0, np.cos(np.deg2rad(ang)), np.sin(np.deg2rad(ang))
            if do_scale:
                offset = opt("offset", 30)
                r_y = np.random.randint(n_y - offset, n_y + 1)          # np.random.random_integers(lo, hi) is inclusive
                p_x = np.random.randint(0, n_x - r_y + 1)
                p_y = np.random.randint(0, n_y - r_y + 1)
                out[ii, 3], out[ii, 4], out[ii, 5], out[ii, 6] = 1.0, p_x, p_y, r_y
        flips = 0
        if do_lr and np.random.randint(max(2, nth)) == 0:
            flips |= 1
        if do_ud and np.random.randint(max(2, nth)) == 0:
            flips |= 2
        out[ii, 7] = flips
    return out


class BatchProvider:
    def __init__(self, X, y, indices, add_dummy_dimension=False, device=None, **kwargs):
        self.X, self.y = X, y
        self.indices = np.asarray(indices)
        self.unused_indices = self.indices.copy()
        self.add_dummy_dimension = add_dummy_dimension
        self.num_labels_per_subject = kwargs.get("num_labels_per_subject", 1)
        if self.num_labels_per_subject > 1:
            self.annotator_range = kwargs.get("annotator_range", range(self.num_labels_per_subject))
        if kwargs.get("resize_to", None):
            raise NotImplementedError("resize_to (UZH prostate loader) is not part of the native input pipeline")
        self.do_augmentations = kwargs.get("do_augmentations", False)
        self.augmentation_options = kwargs.get("augmentation_options", None) or {}
        self.rescale_range, self.rescale_rgb = kwargs.get("rescale_range", None), kwargs.get("rescale_rgb", None)
        if self.rescale_range is not None or self.rescale_rgb:
            raise NotImplementedError("rescale_range / rescale_rgb are not used by the LIDC experiments")
        self.device = torch.device(device) if device is not None else \
            (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self._dev = None

    # ------------------------------------------------------------------ sampling (host, reference order of RNG draws)
    def _draw_indices(self, batch_size):
        if len(self.unused_indices) < batch_size:
            self.unused_indices = self.indices
        batch_indices = np.random.choice(self.unused_indices, batch_size, replace=False)
        self.unused_indices = np.setdiff1d(self.unused_indices, batch_indices)
        return np.sort(batch_indices)                              # the reference sorts for HDF5 (:56-57); order kept

    def _draw_annotators(self, n):
        if self.num_labels_per_subject > 1:
            return np.array([np.random.choice(self.annotator_range) for _ in range(n)], np.int32)
        return np.zeros(n, np.int32)

    def _draw(self, batch_indices):
        ann = self._draw_annotators(len(batch_indices))
        H, W = self.X.shape[1], self.X.shape[2]
        if self.do_augmentations:
            prm = draw_augmentation(len(batch_indices), (H, W), self.augmentation_options)
        else:
            prm = np.zeros((len(batch_indices), 8), np.float32)
            prm[:, 1] = 1.0
        return ann, prm

    # ------------------------------------------------------------------ device residency
    def _resident(self):
        if self._dev is None:
            if self.device.type != "cuda":
                raise _ffi.UzError("no GPU visible: the native input pipeline assembles batches on the device (no CPU fallback)")
            X = torch.as_tensor(np.ascontiguousarray(np.asarray(self.X, dtype=np.float32)))
            y = np.asarray(self.y)
            if y.ndim == 3:
                y = y[..., None]
            self._dev = (X.to(self.device), torch.as_tensor(np.ascontiguousarray(y.astype(np.uint8))).to(self.device))
        return self._dev

    def _assemble(self, batch_indices, ann, prm):
        Xd, yd = self._resident()
        B, H, W, A = len(batch_indices), Xd.shape[1], Xd.shape[2], yd.shape[3]
        tab = torch.from_numpy(np.concatenate([batch_indices.astype(np.int32), ann.astype(np.int32)])).to(self.device, non_blocking=True)
        pd = torch.from_numpy(prm).to(self.device, non_blocking=True)
        x = torch.empty(B, 1, H, W, device=self.device)
        s = torch.empty(B, H, W, device=self.device)
        nl = int(self.augmentation_options.get("nlabels", 0) or max(2, int(yd.max()) + 1 if not self.do_augmentations else 2))
        st = torch.cuda.current_stream(self.device).cuda_stream
        _ffi.check(_ffi.lib().uz_augment_batch(Xd.data_ptr(), yd.data_ptr(), H, W, A, tab.data_ptr(), tab.data_ptr() + 4 * B,
                                               pd.data_ptr(), B, nl, x.data_ptr(), s.data_ptr(), C.c_void_p(st)), "augment_batch")
        return (x if self.add_dummy_dimension else x[:, 0]), s

    # ------------------------------------------------------------------ reference API
    def next_batch(self, batch_size, host=False):
        idx = self._draw_indices(batch_size)
        ann, prm = self._draw(idx)
        x, s = self._assemble(idx, ann, prm)
        return (x.cpu().numpy(), s.cpu().numpy()) if host else (x, s)

    def iterate_batches(self, batch_size, shuffle=True, host=False):
        if shuffle:
            np.random.shuffle(self.indices)
        N = self.indices.shape[0]
        for b_i in range(0, N, batch_size):
            idx = np.sort(self.indices[b_i:b_i + batch_size])
            ann, prm = self._draw(idx)
            x, s = self._assemble(idx, ann, prm)
            yield (x.cpu().numpy(), s.cpu().numpy()) if host else (x, s)
