"""GPU: the PHiSeg forward (and the step behind it) with the deep-level chain launch (csrc/chain.hip) against the per-op tape:
every plan buffer both plans hold is compared by name, then the loss terms and the parameter gradients; finally the time of the
forward tape in both forms (eager replay of the tape, hipEvents)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd.models.phiseg import PHISeg
from unet_zoo_amd._plan import View

B = int(os.environ.get("B", "32"))
nf = [32, 64, 128, 192, 192, 192, 192]
torch.manual_seed(0)
ref = PHISeg(1, 2, nf); ref.chain_px = 0; ref.train()
net = PHISeg(1, 2, nf); net.chain_px = 8192; net.train()      # (the chains are off by default: NativeModel.chain_px)
net.load_state_dict(ref.state_dict())
g = torch.Generator(device="cpu").manual_seed(1)
x = (torch.randn(B, 1, 128, 128, generator=g) * 0.25).clamp(-0.5, 0.5).cuda()
yy, xx = torch.meshgrid(torch.arange(128), torch.arange(128), indexing="ij")
mask = (((yy - 64) ** 2 + (xx - 60) ** 2) < 20 ** 2).float().expand(B, 1, 128, 128).contiguous().cuda()
shapes = [(B, 2, 2 << k, 2 << k) for k in range(5)]
eps = [torch.randn(*s, generator=g).cuda() for s in shapes + shapes]

outs = {}
for name, m in (("ref", ref), ("chain", net)):
    s = m.forward(x, mask, training=True, eps=eps)
    loss = m.loss(mask)
    loss.backward()
    torch.cuda.synchronize()
    outs[name] = (m._cur, float(loss), [t.clone() for t in s])
pr, pc = outs["ref"][0], outs["chain"][0]
print("chain_info", pc.chain_info, "status", pc.chain_status(net._stream()))
print("loss ref %.8g chain %.8g" % (outs["ref"][1], outs["chain"][1]))
byname = {b.name: b for b in pr.bufs}
worst = []
for b in pc.bufs:
    r = byname.get(b.name)
    if r is None or b.packed or r.packed or b.b16 or b.name.startswith("grad:") or ":dy" in b.name or "slab" in b.name or "scratch" in b.name:
        continue
    if (b.N, b.C, b.H, b.W) != (r.N, r.C, r.H, r.W) or b.alias is not None:
        continue
    tc, tr = pc.tensor(View(b)), pr.tensor(View(r))
    if not torch.isfinite(tc).all():
        worst.append((float("inf"), b.name)); continue
    d = (tc - tr).abs().max().item()
    sc = tr.abs().max().item() + 1e-30
    worst.append((d / sc, b.name, d, sc))
worst.sort(reverse=True)
print("buffers compared", len(worst))
for w in worst[:12]:
    print("  ", w)
# backward chains: the outputs of every sub-op, in phase order, against the per-op tape's buffers of the same name
from unet_zoo_amd._plan import _ScratchView
OUT = {"UZ_CH_BN_BWD": [4], "UZ_CH_CONV3": [3], "UZ_CH_CONV3_SMALL_BWD_DATA": [2], "UZ_CH_AVGPOOL_BWD": [1], "UZ_CH_BILINEAR_BWD": [1],
       "UZ_CH_LATENT_HEADS_BWD": [5, 6, 9], "UZ_CH_SLAB_SUM": [1]}
shown = 0
for ch in pc._chains:
    if ch["which"] != "bwd":
        continue
    for e in ch["sub"]:
        for j in OUT.get(e["code"], []):
            v = e["p"][j]
            if not isinstance(v, View):
                continue
            r = byname.get(v.buf.name)
            if r is None:
                continue
            tc, tr = pc.tensor(v), pr.tensor(View(r, v.c0, v.C))
            d = (tc - tr).abs().max().item(); sc = tr.abs().max().item() + 1e-30
            if d / sc > 1e-3 and shown < 25:
                shown += 1
                print("   BWD MISMATCH", ch["net"], "phase", e["level"], e["code"], e["i"], v.buf.name, (v.c0, v.C), "rel", d / sc, "ref max", sc)
gr = ref._ptab.gflat; gc = net._ptab.gflat
print("grad rel err (flat, inf-norm / max)", ((gr - gc).abs().max() / gr.abs().max()).item(), "l2 rel", ((gr - gc).norm() / gr.norm()).item())
for k in range(5):
    d = (outs["ref"][2][k] - outs["chain"][2][k]).abs().max().item()
    print("logits level", k, "max abs diff", d)

def time_fwd(m, n=20):
    plan = m._cur
    for _ in range(3):
        plan.run("fwd", m._stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        plan.run("fwd", m._stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("eager forward tape ms: per-op %.3f  chain %.3f" % (time_fwd(ref), time_fwd(net)))
for m, nm in ((ref, "per-op"), (net, "chain")):
    m.enable_graphs(True)
    for _ in range(3):
        m.forward(x, mask, training=True, eps=eps); l = m.loss(mask); l.backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m.forward(x, mask, training=True, eps=eps); l = m.loss(mask); l.backward()
    torch.cuda.synchronize()
    print(nm, "lane replay fwd+loss+bwd ms/step %.3f" % ((time.perf_counter() - t0) * 100))
print("status", pc.chain_status(net._stream()))
