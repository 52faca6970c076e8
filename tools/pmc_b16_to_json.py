#!/usr/bin/env python3
"""tools/pmc_summary.py outputs of the FETCH_SIZE / WRITE_SIZE passes over tools/pmc_traffic_b16.py -> profiles/r<N>_pmc_traffic_b16.json
(HBM bytes per launch of the bf16-storage kernels on the heaviest PHiSeg3D layer, FETCH_SIZE calibrated on channel_sum_partial as
MI355X_MICROARCH.md prescribes).  usage: pmc_b16_to_json.py <round> <fetch_summary.txt> <write_summary.txt>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KB = 1024


def parse(path):
    out, cur = {}, None
    for ln in open(path):
        m = re.match(r"^(\S.*?)\s+dispatches:", ln)
        if m:
            cur = m.group(1).strip(); out[cur] = {}
            continue
        m = re.match(r"^\s+(\S+)\s+(\d+)", ln)
        if m and cur:
            out[cur][m.group(1)] = float(m.group(2))
    return out


rnd, fetch, write = sys.argv[1], parse(sys.argv[2]), parse(sys.argv[3])
C, Cout, D, H, W = 96, 96, 128, 128, 64
cal_expected = 32 * 32 * 128 * 128 * 4
cal = [v["FETCH_SIZE"] for k, v in fetch.items() if k.startswith("channel_sum_partial")][0] * KB
f = cal_expected / cal
t = D * C * H * W * 2
alg = {"conv_b16_kernel_2_512_32": ("3x3x3 forward / data gradient (mean of the two launches)", 2 * t + Cout * C * 27 * 4),
       "conv_b16_db_kernel_2_512_32": ("3x3x3 forward / data gradient, 64-channel tile, two LDS images (mean of the two launches)", 2 * t + Cout * C * 27 * 4),
       "conv_b16_db_kernel_4_512_32": ("3x3x3 forward / data gradient, 128-channel tile (round 5; mean of the two launches)", 2 * t + Cout * C * 27 * 4),
       "wgrad_split_kernel<32, 64, 1, 2, 2>": ("weight gradient (slabs)", 2 * t),
       "bn_apply_st": ("BatchNorm apply (y -> a)", 2 * t), "bn_bwd_reduce_partial_st": ("BatchNorm backward sums (dA, y)", 2 * t),
       "bn_bwd_apply_st": ("BatchNorm backward apply (dA, y -> dy)", 3 * t)}
d = dict(source="rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (two passes, --kernel-trace only) -- python tools/pmc_traffic_b16.py (tools/prof_b16.sh)",
         layer="Conv3d 96 -> 96 as depth window 288 -> 96, 128 slices of 128 x 64, every tensor in bf16 storage (heaviest layer of PHiSeg3D 5/5 on 4 x 128 x 128 x 64)",
         calibration=dict(kernel="channel_sum_partial (one coalesced dword per lane)", known_read_bytes=cal_expected, FETCH_SIZE_bytes=int(cal), factor=round(f, 4)), kernels={})
for k in fetch:
    if "FETCH_SIZE" not in fetch[k]:
        continue
    for key, (what, a) in alg.items():
        if k.startswith(key):
            fb, wb = int(fetch[k]["FETCH_SIZE"] * KB * f), int(write.get(k, {}).get("WRITE_SIZE", 0) * KB)
            d["kernels"][key] = dict(what=what, hbm_read_bytes=fb, hbm_write_bytes=wb, hbm_bytes=fb + wb, algorithmic_bytes=a, ratio=round((fb + wb) / a, 3))
json.dump(d, open(os.path.join(ROOT, "profiles", "r%s_pmc_traffic_b16.json" % rnd), "w"), indent=1)
print(json.dumps(d, indent=1)[:2500])
