"""Validation metrics: the numpy oracle against golden values produced by the reference's own
utils.generalised_energy_distance / variance_ncc_dist (CPU tier), and the device implementation against both
(GPU tier).  GED is integer pair counting -> exact up to the final float division."""
import numpy as np
import pytest
import torch

from oracle import metrics as OM
from tests import _golden as G

CASES = [0, 1, 2]


@pytest.mark.parametrize("case", CASES)
def test_oracle_metrics_match_reference(case):
    arrays, meta = G.load("metrics")
    s, g, soft = arrays[f"m{case}_samples"].astype(np.int64), arrays[f"m{case}_gts"].astype(np.int64), arrays[f"m{case}_soft"]
    ged = OM.generalised_energy_distance(s, g, nlabels=1, label_range=range(1, 2))
    assert abs(ged - meta[f"m{case}_ged"]) <= 1e-12
    onehot = np.stack([(g == k) for k in range(2)], axis=1).astype(np.int64)
    ncc = float(np.asarray(OM.variance_ncc_dist(soft, onehot)).reshape(-1)[0])
    assert abs(ncc - meta[f"m{case}_ncc"]) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_device_metrics_match_reference(case):
    from unet_zoo_amd import metrics as DM
    dev = torch.device("cuda", 0)
    arrays, meta = G.load("metrics")
    s = torch.from_numpy(arrays[f"m{case}_samples"].astype(np.int64)).to(dev)
    g = torch.from_numpy(arrays[f"m{case}_gts"].astype(np.int64)).to(dev)
    soft = torch.from_numpy(arrays[f"m{case}_soft"]).to(dev)
    ged = DM.generalised_energy_distance(s, g, nlabels=1, label_range=range(1, 2))
    assert abs(ged - meta[f"m{case}_ged"]) <= 1e-12
    onehot = torch.stack([(g == k) for k in range(2)], dim=1).long()
    ncc = DM.variance_ncc_dist(soft, onehot)
    assert abs(ncc - meta[f"m{case}_ncc"]) <= 1e-5
    # pair counts are exact integers
    c = DM.pair_counts(s, g, 1).cpu().numpy()
    sn, gn = arrays[f"m{case}_samples"], arrays[f"m{case}_gts"]
    for i in range(sn.shape[0]):
        for j in range(gn.shape[0]):
            assert c[i, j, 0] == np.count_nonzero((sn[i] == 1) & (gn[j] == 1))
            assert c[i, j, 1] == np.count_nonzero(sn[i] == 1) and c[i, j, 2] == np.count_nonzero(gn[j] == 1)
    # Dice of the first sample against the first ground truth
    d = DM.per_label_dice(s[0], g[0], 2)
    assert np.allclose(d, OM.per_label_dice(sn[0].astype(np.int64), gn[0].astype(np.int64), 2), atol=1e-12)
