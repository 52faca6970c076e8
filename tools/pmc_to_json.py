#!/usr/bin/env python3
"""Merge `tools/pmc_summary.py` outputs of the FETCH_SIZE / WRITE_SIZE passes (tools/pmc_run.sh) into
profiles/r1_pmc_traffic.json.  usage: pmc_to_json.py <fetch_summary.txt> <write_summary.txt>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KB = 1024


def parse(path):
    out, cur = {}, None
    for ln in open(path):
        m = re.match(r"^(\S.*?)\s+dispatches:", ln)
        if m:
            cur = m.group(1).strip()
            continue
        m = re.match(r"^\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)", ln)
        if m and cur:
            out[cur] = float(m.group(2))
    return out


fetch, write = parse(sys.argv[1]), parse(sys.argv[2])
path = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
d = json.load(open(path))
cal_expected = 32 * 128 * 128 * 128 * 4
cal = [v for k, v in fetch.items() if k.startswith("channel_sum_partial")][0] * KB
f = cal_expected / cal
d["calibration"].update(FETCH_SIZE_bytes=int(cal), factor=round(f, 4))
alg_x, alg_y, alg_w = 32 * 224 * 128 * 128 * 4, 32 * 128 * 128 * 128 * 4, 128 * 224 * 9 * 4
names = {"conv_split_kernel<2>": None, "conv_mfma_kernel<3, 2, 2, false>": "conv_mfma_kernel<3,2,2,false> (forward)",
         "conv_mfma_kernel<3, 2, 2, true>": "conv_mfma_kernel<3,2,2,true> (data gradient)",
         "wgrad_fast_kernel<2, 2, 4, 32>": "wgrad_fast_kernel<2,2,4,32> (weight gradient, slabs)"}
for k in fetch:
    if k not in names:
        continue
    fb, wb = int(fetch[k] * KB * f), int(write.get(k, 0) * KB)
    if k == "conv_split_kernel<2>":
        # forward and data gradient share the kernel: the summary is the mean of both launches
        d["kernels"]["conv_split_kernel<2> (forward)"] = dict(
            hbm_read_bytes=fb, hbm_write_bytes=wb, hbm_bytes=fb + wb, algorithmic_bytes=alg_x + alg_w + alg_y,
            ratio=round((fb + wb) / (alg_x + alg_w + alg_y), 3),
            note="mean over the forward and data-gradient launches of the layer (same kernel, same algorithmic bytes)")
    else:
        alg = alg_x + alg_y if "wgrad" in k else alg_x + alg_w + alg_y
        d["kernels"][names[k]] = dict(hbm_read_bytes=fb, hbm_write_bytes=wb, hbm_bytes=fb + wb, algorithmic_bytes=alg, ratio=round((fb + wb) / alg, 3))
json.dump(d, open(path, "w"), indent=1)
print(json.dumps(d["kernels"], indent=1))
