"""Validation metrics of the reference harness, computed on the device.

Same call signatures as the reference helpers (``utils.generalised_energy_distance`` utils.py:148-200,
``utils.variance_ncc_dist`` utils.py:202-247, the per-label Dice of ``UNetModel.validate``
train_model.py:212-224); the pixel loops run in libuz_hip.so (integer pair counts, cross-entropy maps,
correlations), only the O(N*M) scalar bookkeeping stays on the host.
"""
import ctypes as C

import torch

from . import _ffi


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def pair_counts(a, b, label):
    """a (Na,H,W), b (Nb,H,W) integer label maps on the GPU -> int32 (Na, Nb, 3): intersection, |a|, |b| for `label`."""
    a8 = a.reshape(a.shape[0], -1).to(torch.uint8).contiguous()
    b8 = b.reshape(b.shape[0], -1).to(torch.uint8).contiguous()
    out = torch.empty(a8.shape[0], b8.shape[0], 3, dtype=torch.int32, device=a.device)
    _ffi.check(_ffi.lib().uz_label_pair_counts(a8.data_ptr(), a8.shape[0], b8.data_ptr(), b8.shape[0], a8.shape[1], int(label),
                                               out.data_ptr(), _stream(a)), "label_pair_counts")
    return out


def _iou_dist_sum(a, b, label_range, nlabels):
    """sum over all pairs of 1 - mean_label IoU, with the reference's empty-mask conventions."""
    tot = None
    for lbl in label_range:
        c = pair_counts(a, b, lbl).cpu().double()
        inter, ca, cb = c[..., 0], c[..., 1], c[..., 2]
        union = ca + cb - inter
        iou = torch.where((ca == 0) & (cb == 0), torch.ones_like(inter),
                          torch.where((ca == 0) | (cb == 0), torch.zeros_like(inter), inter / union.clamp(min=1)))
        tot = iou if tot is None else tot + iou
    return float((1 - tot / nlabels).sum())


def generalised_energy_distance(sample_arr, gt_arr, nlabels=1, **kwargs):
    label_range = kwargs.get("label_range", range(nlabels))
    N, M = sample_arr.shape[0], gt_arr.shape[0]
    d_sy = _iou_dist_sum(sample_arr, gt_arr, label_range, nlabels)
    d_ss = _iou_dist_sum(sample_arr, sample_arr, label_range, nlabels)
    d_yy = _iou_dist_sum(gt_arr, gt_arr, label_range, nlabels)
    return (2. / (N * M)) * d_sy - (1. / N ** 2) * d_ss - (1. / M ** 2) * d_yy


def variance_ncc_dist(sample_arr, gt_arr):
    """sample_arr (N,K,H,W) softmax samples, gt_arr (M,K,H,W) one-hot (any dtype); returns the mean NCC."""
    N, K, H, W = sample_arr.shape
    M = gt_arr.shape[0]
    s = sample_arr.float().contiguous()
    g = gt_arr.float().contiguous()
    ess = torch.empty(H * W, device=s.device)
    esy = torch.empty(M, H * W, device=s.device)
    out = torch.empty(M, device=s.device)
    L = _ffi.lib()
    _ffi.check(L.uz_ncc_maps(s.data_ptr(), g.data_ptr(), N, M, K, H * W, ess.data_ptr(), esy.data_ptr(), _stream(s)), "ncc_maps")
    _ffi.check(L.uz_ncc(ess.data_ptr(), esy.data_ptr(), M, H * W, out.data_ptr(), _stream(s)), "ncc")
    return float(out.double().mean())


def per_label_dice(pred, gt, n_classes):
    out = []
    for lbl in range(n_classes):
        c = pair_counts(pred.reshape(1, *pred.shape[-2:]), gt.reshape(1, *gt.shape[-2:]), lbl).cpu()[0, 0]
        inter, ca, cb = int(c[0]), int(c[1]), int(c[2])
        if ca == 0 and cb == 0:
            out.append(1.0)
        elif ca == 0 or cb == 0:
            out.append(0.0)
        else:
            out.append(2.0 * inter / float(ca + cb))
    return out
