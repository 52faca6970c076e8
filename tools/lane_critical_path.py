#!/usr/bin/env python3
"""Realised critical path of a lane-replayed tape: from the per-op end times of tools/lane_trace.py (JSON) and the plan's DAG (lane order +
cross-lane waits), walk back from the last op through whichever predecessor finished last.  Prints the chain with each op's duration
(start = the binding predecessor's end) and a per-op-code summary: where a pass's wall time goes, op by op.
usage: python tools/lane_critical_path.py <trace.json> fwd|bwd      (CPU only; traces written since the schedule travels inside them need nothing else)"""
import json, os, sys
rows, which = json.load(open(sys.argv[1])), sys.argv[2]
if "wait" in rows[0]:                    # the trace carries its schedule (lane + cross-lane waits per op): nothing to rebuild
    lanes, waits = [r["lane"] for r in rows], [r["wait"] for r in rows]
else:                                    # older traces: rebuild the plan here with the env the trace was taken under
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.setdefault("UZ_REPLAY", "lanes")
    from unet_zoo_amd.models.phiseg import PHISeg
    net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], device="cpu"); net.train()
    plan = net._build(32, 128, 128, True, True)
    ops = plan.fwd_ops if which == "fwd" else plan.bwd_ops
    sc = plan.scheds[which]
    assert len(rows) == len(ops) and all(o["lane"] == r["lane"] for o, r in zip(ops, rows)), "the trace was taken under another schedule"
    lanes, waits = [o["lane"] for o in ops], [[sc[k].wait[w] for w in range(sc[k].n_wait)] for k in range(len(ops))]
n = len(rows)
end = [r["end_us"] for r in rows]; pred = [None] * n; start = [0.0] * n; last = {}
for k in range(n):
    cands = ([last[lanes[k]]] if lanes[k] in last else []) + list(waits[k])
    if cands:
        b = max(cands, key=lambda j: end[j]); pred[k] = b; start[k] = min(end[b], end[k])
    last[lanes[k]] = k
k = max(range(n), key=lambda j: end[j]); path = []
while k is not None:
    path.append(k); k = pred[k]
path.reverse()
# lane occupancy: a lane is busy from an op's start (its binding predecessor's end) to its end; the rest of the wall it waits
busy = {}
for k in range(n):
    busy[lanes[k]] = busy.get(lanes[k], 0.0) + end[k] - start[k]
print("lane busy us:", {l: round(v) for l, v in sorted(busy.items())}, "of wall", round(max(end)))
agg = {}
print(f"{which}: {n} ops, wall {max(end):.0f} us, realised critical path {len(path)} ops")
for k in path:
    r = rows[k]; d = end[k] - start[k]
    print(f"{k:4d} lane{r['lane']} start {start[k]:8.1f} dur {d:7.1f} {r['code']:24s} {r['shape']}")
    a = agg.setdefault(r["code"], [0, 0.0]); a[0] += 1; a[1] += d
print({c: (n, round(us)) for c, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])})
