"""GPU: per-parameter gradient norms of the headline PHiSeg step against the real reference's digest (tests/golden/phiseg_full_b32_digest),
for the plan as configured (UZ_CHAIN / UZ_CHAIN_BWD): the worst tensors first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import _golden as G
from tests.test_phiseg_gpu import _model, _inputs
fixture = sys.argv[1] if len(sys.argv) > 1 else "phiseg_full_b32_digest"
arrays, meta = G.load(fixture)
net, _ = _model(meta)
net.train()
x, mask, eps = _inputs(meta, 0)
s = net.forward(x, mask, training=True, eps=eps)
loss = net.loss(mask)
loss.backward()
torch.cuda.synchronize()
st = meta["steps"][0]
print("chain", getattr(net._cur, "chain_info", None), "status", net._cur.chain_status(net._stream()))
print("loss", float(loss.detach()), st["loss"])
noise = G.bn_shadowed_biases(st["grad_norms"].keys())
params = dict(net.named_parameters())
rows = []
for k, n in st["grad_norms"].items():
    if k in noise:
        continue
    mine = float(params[k].grad.double().norm())
    rows.append((abs(mine - n) / max(n, 1e-3), k, mine, n))
rows.sort(reverse=True)
for r in rows[:int(os.environ.get("TOP", "25"))]:
    print("  %.3e  %-80s mine %.6g ref %.6g" % r)
import statistics
print("median rel err", statistics.median(r[0] for r in rows), "n", len(rows), "over 1e-2:", sum(r[0] > 1e-2 for r in rows))
