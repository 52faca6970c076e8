#!/usr/bin/env python3
"""Merge the `tools/pmc_summary.py` outputs of the FETCH_SIZE / WRITE_SIZE / MFMA-busy passes (tools/prof_round.sh) into
profiles/r<N>_pmc_traffic.json.  usage: pmc_to_json.py <round> <fetch_summary.txt> <write_summary.txt> <mfma_summary.txt>
FETCH_SIZE is calibrated on `channel_sum_partial` (one coalesced dword per lane over a known byte count), as
MI355X_MICROARCH.md (HBM section) prescribes: gfx950 tallies 128-B requests at 64 B."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KB = 1024


def parse(path):
    out, cur = {}, None
    for ln in open(path):
        m = re.match(r"^(\S.*?)\s+dispatches:", ln)
        if m:
            cur = m.group(1).strip()
            out[cur] = {}
            continue
        m = re.match(r"^\s+(\S+)\s+(\d+)", ln)
        if m and cur:
            out[cur][m.group(1)] = float(m.group(2))
    return out


rnd, fetch, write, mfma = sys.argv[1], parse(sys.argv[2]), parse(sys.argv[3]), parse(sys.argv[4])
N, Cin, Cout, H = 32, 224, 128, 128
cal_expected = N * Cout * H * H * 4
cal = [v["FETCH_SIZE"] for k, v in fetch.items() if k.startswith("channel_sum_partial")][0] * KB
f = cal_expected / cal
alg_x, alg_y, alg_w = N * Cin * H * H * 4, N * Cout * H * H * 4, Cout * Cin * 9 * 4
d = dict(source="rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (three passes, --kernel-trace "
                "only) -- python tools/pmc_traffic.py (tools/prof_round.sh); summaries: profiles/r%s_pmc_*_summary.txt (FETCH/WRITE in KB, "
                "mean per dispatch)" % rnd,
         layer="3x3 conv 224 -> 128 channels, N=32, 128x128 (heaviest layer of PHiSeg 7/5 at batch 32)",
         calibration=dict(kernel="channel_sum_partial (one coalesced dword per lane)", known_read_bytes=cal_expected,
                          FETCH_SIZE_bytes=int(cal), factor=round(f, 4)),
         kernels={}, mfma_busy={})
for k in fetch:
    if "FETCH_SIZE" not in fetch[k]:
        continue
    fb, wb = int(fetch[k]["FETCH_SIZE"] * KB * f), int(write.get(k, {}).get("WRITE_SIZE", 0) * KB)
    if k.startswith("conv_split_kernel<2") or k.startswith("conv_split_kernel_2_512_32"):
        name, alg = "conv_split_kernel<2> f16 (forward)", alg_x + alg_w + alg_y
        note = "mean over the forward and data-gradient launches of the layer (same kernel, same algorithmic bytes)"
    elif k.startswith("wgrad_split_kernel<32"):
        name, alg, note = "wgrad_split_kernel<32,64> f16 (weight gradient, slabs)", alg_x + alg_y, None
    elif k.startswith("bn_") or k.startswith("void bn_"):
        name, alg, note = k, None, None
    else:
        continue
    e = dict(hbm_read_bytes=fb, hbm_write_bytes=wb, hbm_bytes=fb + wb)
    if alg:
        e.update(algorithmic_bytes=alg, ratio=round((fb + wb) / alg, 3))
    if note:
        e["note"] = note
    d["kernels"][name] = e
for k, v in mfma.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v and ("split" in k or "mfma" in k or "wgrad" in k):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA_BUSY over the 4 SIMDs x 256 CUs
        d["mfma_busy"][k] = dict(SQ_VALU_MFMA_BUSY_CYCLES=v["SQ_VALU_MFMA_BUSY_CYCLES"], GRBM_GUI_ACTIVE=v["GRBM_GUI_ACTIVE"],
                                 mfma_busy_fraction=round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024), 3))
path = os.path.join(ROOT, "profiles", "r%s_pmc_traffic.json" % rnd)
json.dump(d, open(path, "w"), indent=1)
print(json.dumps(d, indent=1)[:3000])
