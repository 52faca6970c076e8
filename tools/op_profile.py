#!/usr/bin/env python3
"""usage: op_profile.py [batch] [phiseg|unet|probunet].  Per-op timing of one training step (HIP events around each tape op): which layers cost what."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import FILTERS7 as FILTERS, conv_flops as _cf, op_bytes as _ob, FAMILY
conv_flops = lambda o, _=None: _cf(o)
op_bytes = lambda o, _=None: _ob(o)
from unet_zoo_amd import _ffi
from unet_zoo_amd.models.phiseg import PHISeg
from unet_zoo_amd.synthetic import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MODEL = sys.argv[2] if len(sys.argv) > 2 else "phiseg"
BURST = int(os.environ.get("UZ_OP_PROFILE_BURST", "1"))       # launches per timed bracket (bench.py's family timing uses 4)
import bench
net = bench.build(MODEL); net.train()
x, m, _ = synthetic_batch(B)
x, m = torch.from_numpy(x).cuda(), torch.from_numpy(m).cuda()
for _ in range(2):
    if MODEL == "unet":
        net.forward(x)
    else:
        net.forward(x, m)
    l = net.loss(m); l.backward()
plan = net._cur
L = _ffi.lib(); st = C.c_void_p(net._stream())
rows = []
for which, ops in (("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops)):
    arr, n = plan.tapes[which]
    for k in range(n):
        one = (type(arr[0]) * BURST)(*([arr[k]] * BURST)); best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _ffi.check(L.uz_run_tape(one, BURST, st), "op"); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / BURST)
        o = ops[k]
        rows.append((best, which, o["code"].replace("UZ_OP_", ""), o["i"][:9], conv_flops(o, None), op_bytes(o, plan)))
if os.environ.get("UZ_OP_PROFILE_JSON"):      # every op in tape order with its isolated time: input of tools/critical_path.py
    import json
    json.dump([dict(ms=r[0], tape=r[1], code=r[2], i=list(r[3])) for r in rows], open(os.environ["UZ_OP_PROFILE_JSON"], "w"))
tot = sum(r[0] for r in rows)
print(f"total {tot:.2f} ms over {len(rows)} ops")
if os.environ.get("UZ_OP_PROFILE_STREAMING"):                 # the bandwidth-bound launches (>= 32 MB), slowest rate first
    big = [r for r in rows if not r[4] and r[5] >= bench.LARGE_OP_BYTES]
    for r in sorted(big, key=lambda r: r[5] / r[0]):
        print(f"   {r[2]:14s} {str(list(r[3])):52s} {r[0] * 1e3:7.1f} us {r[5] / 1e6:7.1f} MB {r[5] / r[0] / 1e6:6.0f} GB/s")
agg = {}
for r in rows:
    key = (r[2], tuple(r[3]))
    a = agg.setdefault(key, [0.0, 0, r[4], r[5]]); a[0] += r[0]; a[1] += 1
for (code, i), (ms, cnt, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    extra = f"{fl / (ms / cnt) / 1e9:7.1f} TF/s" if fl else (f"{by / (ms / cnt) / 1e6:7.0f} GB/s" if by else "")
    print(f"{ms:8.3f} ms  x{cnt:<3d} {ms / cnt * 1e3:9.1f} us  {code:18s} {list(i)}  {extra}")
# ---- aggregate by (family, resolution)
byres = {}
for r in rows:
    code, i = r[2], r[3]
    if code.startswith("CONV_FWD") or code.startswith("CONV_BWD"):
        H = i[5]
    elif code.startswith("BN_RELU_FWD"):
        H = i[4]
    elif code.startswith("BN_RELU_BWD"):
        H = i[5]
    else:
        H = i[4] if len(i) > 4 else 0
    fam = FAMILY.get("UZ_OP_" + code, "other")
    d = byres.setdefault((fam, H), [0.0, 0, 0.0]); d[0] += r[0]; d[1] += 1; d[2] += r[4]
print("---- ms by family x resolution")
for fam in sorted({k[0] for k in byres}):
    line = f"{fam:16s}"
    for H in (128, 64, 32, 16, 8, 4, 2):
        d = byres.get((fam, H))
        line += f" | {H:3d}: " + (f"{d[0]:6.2f}ms x{d[1]:<3d}" + (f"{d[2]/d[0]/1e9:5.0f}TF" if d[2] else "       ") if d else " " * 20)
    print(line)
