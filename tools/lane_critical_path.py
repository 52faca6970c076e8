#!/usr/bin/env python3
"""Realised critical path of a lane-replayed tape: from the per-op end times of tools/lane_trace.py (JSON) and the plan's DAG (lane order +
cross-lane waits), walk back from the last op through whichever predecessor finished last.  Prints the chain with each op's duration
(start = the binding predecessor's end) and a per-op-code summary: where a pass's wall time goes, op by op.
usage: python tools/lane_critical_path.py <trace.json> fwd|bwd      (CPU only: the plan is rebuilt here with the env the trace was taken under)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("UZ_REPLAY", "lanes")
from unet_zoo_amd.models.phiseg import PHISeg
rows, which = json.load(open(sys.argv[1])), sys.argv[2]
net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], device="cpu"); net.train()
plan = net._build(32, 128, 128, True, True)
ops = plan.fwd_ops if which == "fwd" else plan.bwd_ops
sc = plan.scheds[which]
assert len(rows) == len(ops) and all(o["lane"] == r["lane"] for o, r in zip(ops, rows)), "the trace was taken under another schedule"
end = [r["end_us"] for r in rows]; pred = [None] * len(ops); start = [0.0] * len(ops); last = {}
for k, o in enumerate(ops):
    cands = ([last[o["lane"]]] if o["lane"] in last else []) + [sc[k].wait[w] for w in range(sc[k].n_wait)]
    if cands:
        b = max(cands, key=lambda j: end[j]); pred[k] = b; start[k] = min(end[b], end[k])
    last[o["lane"]] = k
k = max(range(len(ops)), key=lambda j: end[j]); path = []
while k is not None:
    path.append(k); k = pred[k]
path.reverse()
agg = {}
print(f"{which}: {len(ops)} ops, wall {max(end):.0f} us, realised critical path {len(path)} ops")
for k in path:
    r = rows[k]; d = end[k] - start[k]
    print(f"{k:4d} lane{r['lane']} start {start[k]:8.1f} dur {d:7.1f} {r['code']:24s} {r['shape']}")
    a = agg.setdefault(r["code"], [0, 0.0]); a[0] += 1; a[1] += d
print({c: (n, round(us)) for c, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])})
