"""Which gradient tensors move when the one-launch BatchNorm forward is on (UZ_BN_MID_FWD)?  Small PHiSeg (filters 8/16, B = 4, 128 x 128) against the CPU oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from tests import _golden as G
from unet_zoo_amd.models.phiseg import PHISeg, phiseg_spec
filters, B, HW = [8, 16, 16, 16, 16, 16, 16], 4, 128
sd0 = oracle.deterministic_state_dict(phiseg_spec(1, 2, filters), seed=21)
shapes = oracle.phiseg_eps_shapes(B, HW, HW)
x, mask, eps = oracle.synthetic_batch(B, HW, HW, seed=100, eps_shapes=shapes + shapes)
dev = torch.device("cuda", 0)
net = PHISeg(1, 2, filters, latent_levels=5, image_size=(1, HW, HW)); net.load_state_dict(sd0); net.train()
s = net.forward(torch.from_numpy(x).to(dev), torch.from_numpy(mask).to(dev), training=True, eps=[torch.from_numpy(e).to(dev) for e in eps])
loss = net.loss(torch.from_numpy(mask).to(dev)); loss.backward()
dt = torch.float64
lv = G.leaves({k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in sd0.items()})
e = [torch.from_numpy(a).to(dt) for a in eps]
out = oracle.phiseg_forward(lv, torch.from_numpy(x).to(dt), torch.from_numpy(mask).to(dt), dict(posterior=e[:5], prior=e[5:]))
total, _ = oracle.phiseg_loss(out, torch.from_numpy(mask).to(dt)); total.backward()
print("loss", float(loss), float(total), "logit err", max(float((s[l].cpu().double() - out["s"][l]).abs().max()) for l in range(5)))
noise = G.bn_shadowed_biases(lv.keys())
rows = []
for k, v in lv.items():
    if not v.requires_grad or v.grad is None or k in noise: continue
    mine = dict(net.named_parameters())[k].grad.cpu().double()
    rows.append((float((mine - v.grad).abs().max() / (1e-3 + v.grad.abs().max())), k, float(v.grad.abs().max())))
print("tensors further than 1e-3 from the fp64 oracle, in state_dict order:")
for r in rows:
    if r[0] > 1e-3: print(f"{r[0]:.3e} {r[1]} max|g| {r[2]:.3e}")
print("flags", net.check_bounds())
print("worst", max(rows)[:2], "median", sorted(r[0] for r in rows)[len(rows)//2])
if os.environ.get("UZ_DUMP"):
    plan = net._cur
    out_t = {}
    for b in plan.bufs:
        if "prior.contracting_path.1" in b.name or "prior.pool1" in b.name or "prior.pool2" in b.name:
            from unet_zoo_amd._plan import View
            out_t[b.name] = plan.tensor(View(b)).detach().cpu().clone()
    for k, p_ in net.named_parameters():
        if k.startswith("prior.contracting_path.1") and p_.grad is not None:
            out_t["pgrad:" + k] = p_.grad.detach().cpu().clone()
    torch.save(out_t, os.environ["UZ_DUMP"])
    print("dumped", len(out_t))
