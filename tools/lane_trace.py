#!/usr/bin/env python3
"""Un-profiled device timeline of a lane-replayed PHiSeg step (UZ_REPLAY=lanes): one timing event behind every op.
usage: UZ_REPLAY=lanes python tools/lane_trace.py [fwd|bwd] [out.json]"""
import ctypes as C, json, os, sys
os.environ.setdefault("UZ_REPLAY", "lanes")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_zoo_amd import _ffi
from unet_zoo_amd.models.phiseg import PHISeg
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], latent_levels=5, image_size=(1, 128, 128)).to(dev)
net.train(); net.enable_graphs(True)
x = torch.randn(32, 1, 128, 128, device=dev); m = torch.randint(0, 2, (32, 128, 128), device=dev)
L = _ffi.lib()
def step(trace=None):
    if trace == "fwd": L.uz_lane_trace(1, 1024, None, 0)
    net.forward(x, m)
    out = None
    if trace == "fwd": out = dump()
    loss = net.loss(m)
    if trace == "bwd": L.uz_lane_trace(1, 1024, None, 0)
    loss.backward()
    if trace == "bwd": out = dump()
    return out
def dump():
    buf = (C.c_float * 1024)()
    n = L.uz_lane_trace(0, 0, buf, 1024)
    return [buf[k] for k in range(n)]
for _ in range(6): step()
if int(os.environ.get("UZ_TUNE_SCHEDULE", "0")) > 0:        # trace the profile-guided schedule instead of the static one
    print("tuned:", net.tune_schedule(step, rounds=int(os.environ["UZ_TUNE_SCHEDULE"])))
    step()
torch.cuda.synchronize()
ends = step(which)
plan = next(iter(net._plans.values()))
ops = plan.fwd_ops if which == "fwd" else plan.bwd_ops
sc = plan.scheds[which]
rows = []
last = {}
for k, (o, e) in enumerate(zip(ops, ends)):
    i, c = o["i"], o["code"].replace("UZ_OP_", "")
    shape = ""
    if c.startswith("CONV"): shape = f"{i[0]}->{i[2]}@{i[5]}x{i[6]}k{i[7]}"
    elif c == "BN_RELU_FWD": shape = f"C{i[0]}@{i[4]}x{i[5]}"
    elif c == "BN_RELU_BWD": shape = f"C{i[1]}@{i[5]}x{i[6]}"
    rows.append(dict(k=k, lane=o["lane"], gid=o["gid"], code=c, shape=shape, end_us=round(e * 1e3, 1), since_lane_prev_us=round((e - last.get(o["lane"], 0.0)) * 1e3, 1),
                     wait=[sc[k].wait[w] for w in range(sc[k].n_wait)]))
    last[o["lane"]] = e
print(f"{which}: {len(rows)} ops, last end {max(r['end_us'] for r in rows):.0f} us")
for r in rows: print(f"{r['k']:4d} lane{r['lane']} {r['end_us']:9.1f} (+{r['since_lane_prev_us']:7.1f}) {r['code']:22s} {r['shape']}")
if len(sys.argv) > 2: json.dump(rows, open(sys.argv[2], "w"))
