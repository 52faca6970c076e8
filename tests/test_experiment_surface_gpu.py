"""Every distinct (model, filters, image size, classes, reversible) combination of the reference's 28 experiment files
(models/experiments/*.py; table below transcribed from their `filter_channels / image_size / n_classes / use_reversible`
attributes) takes a training step on the device; the 3-label 192 x 192 UZH shape - 3 x 3 planes at the deepest level - is
compared with the oracle.  Batches are cut down to keep the tier short: the batch only scales the kernels' grid."""
import numpy as np
import pytest
import torch

import oracle
import unet_zoo_amd  # noqa: F401
from tests import _golden as G

NF7 = [32, 64, 128, 192, 192, 192, 192]
BIG = [32, 64, 128, 192, 256, 256, 256]
CASES = [  # (experiment files, filters, H = W, classes, reversible, batch)
    ("phiseg_7_5_{12..56}", NF7, 128, 2, False, 3),
    ("phiseg_rev_7_5_{12..64}", NF7, 128, 2, True, 3),
    ("phiseg_big", BIG, 128, 2, False, 2),
    ("phiseg_big_reversible", BIG, 128, 2, True, 2),
    ("phiseg_uzh_7_5_192", NF7, 192, 3, False, 2),
    ("phiseg_uzh_rev_7_5_192", NF7, 192, 3, True, 2),
    ("phiseg_uzh_7_5_256 / rev", NF7, 256, 3, True, 2),
    ("phiseg_uzh_7_5_384 / rev", NF7, 384, 3, False, 1),
    ("phiseg_uzh_7_5_512 / rev", NF7, 512, 3, True, 1),
]


def _labels(batch, hw, classes, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    x = np.clip(rng.standard_normal((batch, 1, hw, hw)).astype(np.float32) * 0.25, -0.5, 0.5)
    yy, xx = np.mgrid[0:hw, 0:hw]
    m = np.zeros((batch, 1, hw, hw), np.float32)
    for b in range(batch):
        for k in range(1, classes):
            r = hw * 0.3 / k
            m[b, 0][(yy - hw * 0.5) ** 2 + (xx - hw * (0.4 + 0.1 * b)) ** 2 <= r * r] = k
    return x, m


@pytest.mark.gpu
@pytest.mark.parametrize("files,filters,hw,classes,rev,batch", CASES, ids=[c[0] for c in CASES])
def test_experiment_shape_trains_on_the_device(files, filters, hw, classes, rev, batch):
    from unet_zoo_amd.models.phiseg import PHISeg
    from unet_zoo_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    net = PHISeg(1, classes, filters, latent_levels=5, image_size=(1, hw, hw), reversible=rev)
    net.train()
    net.enable_graphs(True)
    opt = FusedAdam(net, lr=1e-4, weight_decay=1e-5)
    x, m = _labels(batch, hw, classes, 7)
    xd, md = torch.from_numpy(x).to(dev), torch.from_numpy(m).to(dev)
    losses = []
    for _ in range(4):                                   # eager, capture, two replays
        s = net.forward(xd, md, training=True)
        loss = net.loss(md)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)), losses
    assert len(s) == 5 and tuple(s[0].shape) == (batch, classes, hw, hw)
    if not rev:                                          # same batch four times: the step must reduce its loss.  (Freshly initialised
        assert losses[-1] < losses[0], losses            # reversible stacks start with activations grown by (1 + gamma) per block and
    #                                                      KL terms of 1e8: their first steps are not monotone - finiteness only.)
    net.eval()
    with torch.no_grad():
        out = net.accumulate_output(net.forward(xd, md, training=False), use_softmax=True)
    assert bool(torch.isfinite(out).all()) and abs(float(out.sum(1).mean()) - 1.0) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("rev", [False, True])
def test_uzh_192_three_labels_vs_oracle(rev):
    """(1, 192, 192), 3 labels (phiseg_uzh_7_5_192.py): planes of 3 x 3 at the deepest level, odd sizes in the pyramid's tail."""
    from unet_zoo_amd.models.phiseg import PHISeg, phiseg_spec
    dev = torch.device("cuda", 0)
    hw, B, K = 192, 2, 3
    sd0 = oracle.deterministic_state_dict(phiseg_spec(1, K, NF7, reversible=rev), seed=41)
    if rev:
        for k, v in sd0.items():
            if k.endswith("convolution.1.weight"):
                sd0[k] = v * 0.3
    net = PHISeg(1, K, NF7, latent_levels=5, image_size=(1, hw, hw), reversible=rev)
    net.load_state_dict(sd0)
    net.train()
    shapes = oracle.phiseg_eps_shapes(B, hw, hw)
    _, _, eps = oracle.synthetic_batch(B, hw, hw, seed=11, eps_shapes=shapes + shapes)
    x, m = _labels(B, hw, K, 5)
    s = net.forward(torch.from_numpy(x).to(dev), torch.from_numpy(m).to(dev), training=True, eps=[torch.from_numpy(e).to(dev) for e in eps])
    loss = net.loss(torch.from_numpy(m).to(dev))
    e = [torch.from_numpy(a) for a in eps]
    with torch.no_grad():
        out = oracle.phiseg_forward(dict(sd0), torch.from_numpy(x), torch.from_numpy(m), dict(posterior=e[:5], prior=e[5:]))
        total, _ = oracle.phiseg_loss(out, torch.from_numpy(m), num_classes=K)
    for l in range(5):
        ref = out["s"][l].numpy()
        assert G.maxabs(s[l].cpu().numpy(), ref) <= 1e-4 * max(1.0, float(np.abs(ref).max())), (rev, l)
    assert abs(float(loss) - float(total)) <= 5e-5 * abs(float(total)), (float(loss), float(total))
