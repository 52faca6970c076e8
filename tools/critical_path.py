#!/usr/bin/env python3
"""Where does a PHiSeg step's wall time come from?  Combines the plan's scheduled DAG (built here, no GPU needed) with the isolated
per-op times of tools/op_profile.py (UZ_OP_PROFILE_JSON, measured on the GPU box with the same build and environment):
  * sum of op times, critical path of the DAG (the bound no lane count can beat), makespan of the schedule as captured
    (lane order + cross-lane waits) with every op at its isolated time;
  * the ops on the critical path, by family.
usage: critical_path.py gpurun_out/r4_op_times.json [batch=32]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_zoo_amd  # noqa: F401
from unet_zoo_amd.models.phiseg import PHISeg

times = json.load(open(sys.argv[1]))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], device="cpu")
net.train()
plan = net._build(B, 128, 128, True, True)
out = {}
for tape, ops in (("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops)):
    t = [r for r in times if r["tape"] == tape]
    assert len(t) == len(ops), (tape, len(t), len(ops))
    for r, o in zip(t, ops):
        assert r["code"] == o["code"].replace("UZ_OP_", "") and r["i"] == o["i"][:9], (r, o["code"], o["i"])
    ms = [r["ms"] for r in t]
    dag = plan.dag[id(ops)]
    G = len(dag)
    cost = [sum(ms[g["first"]:g["last"] + 1]) for g in dag]
    # critical path
    fin, pred = [0.0] * G, [None] * G
    for k, g in enumerate(dag):
        s = 0.0
        for d in g["deps"]:
            if fin[d] > s:
                s, pred[k] = fin[d], d
        fin[k] = s + cost[k]
    end = max(range(G), key=lambda k: fin[k])
    path, k = [], end
    while k is not None:
        path.append(k)
        k = pred[k]
    path.reverse()
    # makespan of the captured schedule: a group starts behind its lane predecessor and its dependencies
    lane_free, fin2 = {}, [0.0] * G
    for k, g in enumerate(dag):
        s = max([lane_free.get(g["lane"], 0.0)] + [fin2[d] for d in g["deps"]])
        fin2[k] = s + cost[k]
        lane_free[g["lane"]] = fin2[k]
    fam = {}
    for k in path:
        for j in range(dag[k]["first"], dag[k]["last"] + 1):
            o = ops[j]
            c = o["code"].replace("UZ_OP_", "")
            H = o["i"][5] if c.startswith("CONV") or c == "BN_RELU_BWD" else (o["i"][4] if len(o["i"]) > 4 else 0)
            key = (c, H)
            fam[key] = fam.get(key, 0.0) + ms[j]
    out[tape] = dict(ops=len(ops), groups=G, sum_ms=round(sum(ms), 3), critical_path_ms=round(fin[end], 3), schedule_makespan_ms=round(max(fin2), 3),
                     path_groups=len(path))
    print(f"{tape}: {len(ops)} ops in {G} groups; sum {sum(ms):.2f} ms; critical path {fin[end]:.2f} ms over {len(path)} groups; "
          f"2-lane schedule at isolated op times {max(fin2):.2f} ms")
    for (c, H), v in sorted(fam.items(), key=lambda kv: -kv[1])[:14]:
        print(f"      on the critical path: {v:6.3f} ms  {c} @ {H}")
print(json.dumps(out))
