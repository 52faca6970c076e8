"""Per-step deviation of the PHiSeg mid fixture trajectory from the reference golden values (loss, logits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import oracle
from tests import _golden as G
from tests.test_phiseg_gpu import _model, _inputs
from unet_zoo_amd.optim import FusedAdam
name = sys.argv[1] if len(sys.argv) > 1 else "phiseg_mid"
arrays, meta = G.load(name)
net, _ = _model(meta)
opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
for step, st in enumerate(meta["steps"]):
    x, mask, eps = _inputs(meta, step)
    net.forward(x, mask, training=True, eps=eps)
    loss = net.loss(mask)
    net.zero_grad(); loss.backward(); opt.step()
    keys = [k for k in arrays if k.startswith(f"step{step}")]
    print(step, "loss rel dev %.3e" % (abs(float(loss) - st["loss"]) / abs(st["loss"])), os.environ.get("UZ_CONV_MATH", "default"))
