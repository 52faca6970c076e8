"""CPU restatement (numpy) of the reference's validation metrics - TEST INFRASTRUCTURE (see oracle/__init__.py).

generalised_energy_distance  utils.py:148-200   (IoU from medpy.metric.jc; MedPy==0.4.0 is pinned in
                                                 requirements.txt:17 but not vendored: jc(a, b) =
                                                 |a & b| / |a | b| on boolean arrays, dc(a, b) = 2|a & b| / (|a| + |b|))
variance_ncc_dist / ncc      utils.py:130-145,202-247
per-label Dice               train_model.py:212-224
"""
import numpy as np


def jc(a, b):
    a, b = np.asarray(a).astype(bool), np.asarray(b).astype(bool)
    return float(np.count_nonzero(a & b)) / float(np.count_nonzero(a | b))


def dc(a, b):
    a, b = np.asarray(a).astype(bool), np.asarray(b).astype(bool)
    inter, sa, sb = np.count_nonzero(a & b), np.count_nonzero(a), np.count_nonzero(b)
    return 2.0 * inter / float(sa + sb) if sa + sb else 0.0


def _dist(m1, m2, label_range, nlabels):
    ious = []
    for lbl in label_range:
        a, b = (m1 == lbl), (m2 == lbl)
        if a.sum() == 0 and b.sum() == 0:
            ious.append(1)
        elif (a.sum() > 0 and b.sum() == 0) or (a.sum() == 0 and b.sum() > 0):
            ious.append(0)
        else:
            ious.append(jc(a, b))
    return 1 - (sum(ious) / nlabels)


def generalised_energy_distance(sample_arr, gt_arr, nlabels=1, label_range=None):
    label_range = range(nlabels) if label_range is None else label_range
    N, M = sample_arr.shape[0], gt_arr.shape[0]
    d_sy = sum(_dist(sample_arr[i], gt_arr[j], label_range, nlabels) for i in range(N) for j in range(M))
    d_ss = sum(_dist(sample_arr[i], sample_arr[j], label_range, nlabels) for i in range(N) for j in range(N))
    d_yy = sum(_dist(gt_arr[i], gt_arr[j], label_range, nlabels) for i in range(M) for j in range(M))
    return (2. / (N * M)) * d_sy - (1. / N ** 2) * d_ss - (1. / M ** 2) * d_yy


def ncc(a, v):
    a, v = a.flatten(), v.flatten()
    a = (a - np.mean(a)) / (np.std(a) * len(a))
    v = (v - np.mean(v)) / np.std(v)
    return np.correlate(a, v)


def variance_ncc_dist(sample_arr, gt_arr):
    """sample_arr (N,K,H,W) float softmax samples, gt_arr (M,K,H,W) integer one-hot."""
    def xent(m_samp, m_gt, eps=1e-8):
        return -1.0 * np.sum(m_gt * np.log(m_samp + eps), axis=0)
    mean_seg = np.mean(sample_arr, axis=0)
    N, M = sample_arr.shape[0], gt_arr.shape[0]
    E_ss = np.mean(np.stack([xent(sample_arr[i], mean_seg) for i in range(N)]).astype(np.float64), axis=0)
    out = []
    for j in range(M):
        E_sy = np.mean(np.stack([xent(sample_arr[i], gt_arr[j]) for i in range(N)]).astype(np.float64), axis=0)
        out.append(ncc(E_ss, E_sy))
    return (1 / M) * sum(out)


def per_label_dice(pred, gt, n_classes):
    out = []
    for lbl in range(n_classes):
        a, b = (pred == lbl), (gt == lbl)
        if b.sum() == 0 and a.sum() == 0:
            out.append(1.0)
        elif (a.sum() > 0 and b.sum() == 0) or (a.sum() == 0 and b.sum() > 0):
            out.append(0.0)
        else:
            out.append(dc(a, b))
    return out
