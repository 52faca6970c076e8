"""GPU: per-phase timeline of the deep-level chain launch (csrc/chain.hip): the 100 MHz stamps workgroup 0 leaves behind every phase,
the launch alone on the chip (hipEvents around the one tape op), and which sub-ops each phase holds."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd.models.phiseg import PHISeg
from unet_zoo_amd import _ffi

B = int(os.environ.get("B", "32"))
which = os.environ.get("TAPE", "fwd")
net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192]); net.chain_px = 8192; net.train()      # (off by default; UZ_CHAIN overrides)
x = (torch.randn(B, 1, 128, 128) * 0.25).clamp(-0.5, 0.5).cuda()
mask = (torch.rand(B, 1, 128, 128) > 0.7).float().cuda()
for _ in range(2):
    net.forward(x, mask, training=True); l = net.loss(mask); l.backward()
torch.cuda.synchronize()
plan = net._cur
L = _ffi.lib()
arr, n = plan.tapes[which]
ops = plan.fwd_ops if which == "fwd" else plan.bwd_ops
k = next(j for j, o in enumerate(ops) if o["code"] == "UZ_OP_CHAIN")
one = (_ffi.uz_op * 1).from_address(C.addressof(arr) + k * C.sizeof(_ffi.uz_op))
st = C.c_void_p(net._stream())
reps = 20
for _ in range(3):
    L.uz_run_tape(one, 1, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    L.uz_run_tape(one, 1, st)
e1.record(); torch.cuda.synchronize()
print("chain launch alone: %.1f us" % (e0.elapsed_time(e1) * 1e3 / reps), plan.chain_info, "status", plan.chain_status(net._stream()))
idx = ops[k]["p"][0][1]
ch = plan._chains[idx]
stt = ch["dev"]["state"].cpu()
stamps = stt[32:].view(torch.int64)
nph = ch["dev"]["n_phases"]
t = [(stamps[j].item() - stamps[0].item()) / 100.0 for j in range(nph)]        # us (100 MHz); the last phase has no barrier behind it
print("phases", nph, "stamped span %.1f us" % t[nph - 1])
cur, rows = None, []
for e in ch["sub"]:
    if e["level"] != cur:
        cur = e["level"]; rows.append([])
    rows[-1].append(e)
prev = 0.0
agg = {}
for ph, es in enumerate(rows):
    end = t[ph + 1] if ph + 1 < nph else float("nan")
    dur = end - prev
    desc = "; ".join(f"{e['code'][6:]}{[e['i'][j] for j in (0, 2, 4, 5, 7)] if e['code'] == 'UZ_CH_CONV3' else e['i'][:5]}x{e['ntiles']}" for e in es)
    kind = "+".join(sorted({e["code"][6:] for e in es}))
    if dur == dur:
        a = agg.setdefault(kind, [0, 0.0]); a[0] += 1; a[1] += dur
    print(f"  phase {ph:3d} {dur:8.1f} us  tiles {sum(e['ntiles'] for e in es):5d}  {desc[:170]}")
    prev = end
print("by kind:")
for kind, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"   {kind:40s} {cnt:3d} phases {us:8.1f} us")
if os.environ.get("UZ_CHAIN_DEBUG_PHASE"):
    dbg = stt[32 + 1024:].view(torch.int64)
    base = dbg[0].item()
    print("in-tile stamps (us since the phase began on the debug workgroup):", {j: round((dbg[j].item() - base) / 100.0, 2) for j in range(32) if dbg[j].item()})
