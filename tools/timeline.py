#!/usr/bin/env python3
"""What bounds a training step?  From a rocprofv3 --kernel-trace CSV of a hipGraph-replayed run: for the last complete step the
wall time split by WHAT is in flight - a device-filling matrix kernel (split / fp32 convolution on >= 16 x 16 planes), only
streaming kernels (BatchNorm, resampling, point-wise), only latency-bound small launches, or nothing.
usage: timeline.py <kernel_trace.csv> [out.json]"""
import csv, json, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))


def wgs(r):
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    w = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    return g // max(1, w)


def klass(name, nwg, dur_us):
    n = name
    if re.search(r"conv_splitp?(_bn|_relu)?_kernel|conv_bf16|wgrad_split_kernel|conv_mfma_kernel|wgrad_kernel|wgrad_fast_kernel", n):
        return "matrix_heavy" if (nwg >= 192 and dur_us >= 25) else "matrix_small"
    if re.search(r"bn_|relu_bwd|bilinear|avgpool|nearest|c1_|adam|pack_all|absmax|zero_k|copy_k|ce_|kl_|latent|add_views|splitk_reduce|wgrad_reduce|chan_|pack_split", n):
        return "streaming" if dur_us >= 12 else "small"
    return "small"


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], wgs(r)) for r in rows)
adam = [k for k, e in enumerate(ev) if "adam" in e[2].lower()]
ends = [k for j, k in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] - k > 5]
step = ev[ends[-2] + 1:ends[-1] + 1]
t0, t1 = step[0][0], max(e[1] for e in step)
pts = []
sums = {}
for s, e, n, w in step:
    k = klass(n, w, (e - s) / 1e3)
    sums[k] = sums.get(k, 0) + (e - s)
    pts += [(s, k, 1), (e, k, -1)]
pts.sort(key=lambda t: (t[0], t[2]))
cnt = {"matrix_heavy": 0, "matrix_small": 0, "streaming": 0, "small": 0}
out = {"idle": 0, "matrix_heavy in flight": 0, "of which a streaming kernel beside it": 0, "only streaming": 0, "only small matrix / small launches": 0}
last = pts[0][0]
for t, k, d in pts:
    dt = t - last
    if cnt["matrix_heavy"]:
        out["matrix_heavy in flight"] += dt
        if cnt["streaming"]:
            out["of which a streaming kernel beside it"] += dt
    elif cnt["streaming"]:
        out["only streaming"] += dt
    elif cnt["matrix_small"] or cnt["small"]:
        out["only small matrix / small launches"] += dt
    else:
        out["idle"] += dt
    cnt[k] += d
    last = t
res = dict(step_wall_ms=round((t1 - t0) / 1e6, 3), kernels=len(step),
           wall_ms_by_what_is_in_flight={k: round(v / 1e6, 3) for k, v in out.items()},
           kernel_time_sum_ms={k: round(v / 1e6, 3) for k, v in sums.items()})
print(json.dumps(res, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
