#!/usr/bin/env python3
"""Headline benchmark: images/sec of one training step (forward + loss + backward + gradient all-reduce + Adam)
at 128x128, batch 32 per GPU, fp32 in / fp32 out, on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5                      # PHiSeg 7/5 = BASELINE.json configs[3] (headline)
    python bench.py --model unet      ...                               # configs[1]: Unet(1,2,[32,64,128,192])
    python bench.py --model probunet  ...                               # configs[2]: ProbabilisticUnet latent 6 (+ 8-sample decode)
    python bench.py --model phiseg3d  ...                               # configs[4]: PHISeg3D 5-level, 4x128x128x64, 1 volume per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement).  Synthetic LIDC-like inputs,
random-init weights, inputs resident in HBM before the timed region.

Extra objects on the line:
  roofline      the DOMINANT KERNEL (longest convolution launch of the step) timed live with HIP events on its launch
                stream: `achieved` = its algorithmic fp32-equivalent FLOPs per launch / average launch duration, `peak` =
                the roof of the pipe it runs on (2500/3 = 833.3 TFLOP/s for the split-fp16 kernels: three fp16 piece
                products per fp32 product; 157.3 for the fp32-MFMA kernels), `traffic` = PMC-measured HBM bytes per launch
                (profiles/*pmc_traffic.json).  `roofline.step` is the step-level view: `achieved` = images/s x the model's
                algorithmic GFLOP/image; `peak` = the BINDING roof of the step - every convolution
                op of the tape priced against the pipe it is routed to, FLOP-weighted harmonically; `frac_vs_fp32_mfma`
                keeps BASELINE.md's figure (achieved / 157.3); `fp32_mfma_only` is the same run with UZ_CONV_MATH=f32
                (like-for-like against 157.3).  `families`: per-family sums, conv families against their binding roof,
                streaming families against 8 TB/s (`large_ops`: the launches big enough to be bandwidth-bound).  With
                --no-profile or more than one GPU the object holds the step-level view only.
  cpu_baseline  the CPU oracle (functional torch restatement of the reference graph = "port") timed on this box's host
                cores on a bounded sample of the same workload (batch stated in `sample`).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")      # see unet-zoo_amd/__init__.py: must be in place before HIP initialises
os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", os.environ["GPU_MAX_HW_QUEUES"] if os.environ["GPU_MAX_HW_QUEUES"] in ("1", "2", "3", "4") else "4")

FILTERS7 = [32, 64, 128, 192, 192, 192, 192]
FILTERS4 = [32, 64, 128, 192]
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md chip-level parameters
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16 / bf16 MFMA peak
SPLIT_PRODUCTS = 3                # fp16 piece products per fp32 product in conv_split.hip / conv_wgrad_split.hip (split_f16.h)
PEAK_SPLIT_TFLOPS = PEAK_F16_MFMA_TFLOPS / SPLIT_PRODUCTS
HBM_PEAK_GBS = 8000.0
LARGE_OP_BYTES = 32e6

# algorithmic work per image (fwd+bwd), unfused tensor bytes per image, fixed bytes per step
MODELS = {
    "phiseg": dict(gflop=100.36, gb_img=1.077, gb_step=0.294, metric="images/sec fwd+bwd PHiSeg-7 128x128 bs32",
                   workload="PHiSeg 7 resolution / 5 latent levels, filters 32-64-128-192x4, 1x128x128, fwd+loss+bwd+Adam (BASELINE configs[3])"),
    "unet": dict(gflop=20.86, gb_img=0.288, gb_step=0.028, metric="images/sec fwd+bwd U-Net-4 128x128 bs32",
                 workload="vanilla U-Net 4-level, filters 32-64-128-192, 1x128x128, fwd+loss+bwd+Adam (BASELINE configs[1])"),
    "probunet": dict(gflop=40.74, gb_img=0.736, gb_step=0.565, metric="images/sec fwd+bwd ProbU-Net(latent 6) 128x128 bs32",
                     workload="Probabilistic U-Net, filters 32-64-128-192x4, latent_dim 6, no_convs_fcomb 3, 1x128x128, "
                              "fwd+loss+bwd+Adam (BASELINE configs[2]); 8 posterior-sample decodes timed separately"),
    # BASELINE configs[4]: one 4x128x128x64 volume per GPU ("batch 8, 8 GPUs"); work per volume is taken from the plan's own
    # convolution ops at run time (BASELINE.md has no row for it).  "%s" = the storage the plan actually chose (bf16 by default, --storage f32 for fp32).
    "phiseg3d": dict(gflop=None, gb_img=None, gb_step=None, metric="volumes/sec fwd+bwd PHiSeg3D-5 4x128x128x64, 1 volume per GPU", unit="volumes/s",
                     workload="PHISeg3D 5 resolution / 5 latent levels, filters 32-64-128-192-192, 4 input channels, 3 labels, one 128x128x64 "
                              "volume per GPU, fwd+loss+bwd+Adam (BASELINE configs[4], %s storage)"),
}
FILTERS3D, DHW3D = [32, 64, 128, 192, 192], (128, 128, 64)


def conv_dims(op):
    """(kind, cin, cout, n, h, w, ks) of a conv-family tape op; kind 0 fwd / 1 dgrad / 2 wgrad."""
    i, c = op["i"], op["code"]
    if c == "UZ_OP_CONV_FWD":
        return 0, i[0], i[2], i[4], i[5], i[6], i[7]
    if c == "UZ_OP_CONV_BWD_DATA":
        return 1, i[2], i[0], i[4], i[5], i[6], i[7]
    if c == "UZ_OP_CONV_BWD_WEIGHT":
        return 2, i[0], i[2], i[4], i[5], i[6], i[7]
    return None


def conv_flops(op):
    d = conv_dims(op)
    if d is None:
        return 0.0
    _, cin, cout, n, h, w, ks = d
    return 2.0 * n * h * w * cin * cout * ks * ks


def _esz(op, bit):
    """Bytes per element of the tensor operand behind format bit `bit` of a plan op (i[13], Plan._b16_pass): 2 in bf16 storage, else 4."""
    i = op["i"]
    return 2.0 if (len(i) > 13 and (i[13] >> bit) & 1) else 4.0


def conv_bytes(op):
    """Algorithmic HBM bytes of a convolution op: input + output tensor once (at their storage width), weights once."""
    d = conv_dims(op)
    if d is None:
        return 0.0
    kind, cin, cout, n, h, w, ks = d
    # format bits: forward (x, y), data gradient (dy, dx), weight gradient (x, dy)
    e_cin, e_cout = {0: (_esz(op, 0), _esz(op, 1)), 1: (_esz(op, 1), _esz(op, 0)), 2: (_esz(op, 0), _esz(op, 1))}[kind]
    win = op["i"][0] == 3 * op["i"][1]                       # depth window of a volume (3 C channels over a C-channel buffer): read once
    if win and kind in (0, 2):
        cin_t, cout_t = cin // 3, cout
    elif win:
        cin_t, cout_t = cin, cout // 3
    else:
        cin_t, cout_t = cin, cout
    return n * h * w * (cin_t * e_cin + cout_t * e_cout) + 4.0 * cin * cout * ks * ks


def conv_roof(op, L):
    """Peak TFLOP/s of the pipe the library routes this op to (uz_conv_route), or None for the streaming 1x1 heads."""
    kind, cin, cout, n, h, w, ks = conv_dims(op)
    r = L.uz_conv_route(kind, cin, cout, n, h, w, ks)
    split_roof = PEAK_F16_MFMA_TFLOPS if L.uz_get_conv_math() == 3 else PEAK_SPLIT_TFLOPS      # bf16 mode: one product per MAC
    return {0: PEAK_F32_MFMA_TFLOPS, 1: split_roof, 2: None}[r]


def _bn_limit(which, H, W):
    from unet_zoo_amd import _ffi
    L = _ffi.lib()
    return (L.uz_bn_fwd_fused_limit if which == "fwd" else L.uz_bn_bwd_fused_limit)(H, W)


def op_bytes(op):
    """Algorithmic HBM bytes of a streaming (non-conv) op: every tensor argument read or written once."""
    c, i = op["code"], op["i"]
    f4 = 4.0
    if c == "UZ_OP_BN_RELU_FWD":
        C, N, H, W, training = i[0], i[3], i[4], i[5], i[6]
        # three passes (statistics: y; apply: y, a) only on the streaming path without the convolution's partials (i[8] > 0); the
        # one-launch paths (channel's batch in registers, up to uz_bn_fwd_fused_limit) read y once
        big = N * H * W > _bn_limit("fwd", H, W) and not (len(i) > 8 and i[8])
        return C * N * H * W * ((2 if (big and training) else 1) * _esz(op, 0) + _esz(op, 1))          # (format bits: y, a)
    if c in ("UZ_OP_BN_RELU_BWD", "UZ_OP_RELU_BWD"):
        C, N, H, W = i[1], i[4], i[5], i[6]
        big = N * H * W > _bn_limit("bwd", H, W)                    # reduce (dA, y) + apply (dA, y, dy); one-launch paths: dA, y, dy
        if c == "UZ_OP_RELU_BWD":
            return f4 * C * N * H * W * 3
        return C * N * H * W * ((2 if big else 1) * (_esz(op, 0) + _esz(op, 1)) + _esz(op, 2))        # (format bits: dA, y, dy)
    if c in ("UZ_OP_AVGPOOL_FWD", "UZ_OP_AVGPOOL_BWD"):
        C, N, H, W = i[0], i[3], i[4], i[5]
        return f4 * C * N * H * W * 1.25
    if c in ("UZ_OP_BILINEAR_FWD", "UZ_OP_BILINEAR_BWD"):
        C, N, H, W = i[0], i[3], i[4], i[5]
        acc = f4 if (c == "UZ_OP_BILINEAR_BWD" and len(i) > 7 and i[7]) else 0                            # dx += ...: the low-resolution gradient is read as well
        return C * N * H * W * (f4 + acc + 4 * _esz(op, 1 if c == "UZ_OP_BILINEAR_FWD" else 0))         # (only the high-resolution side may be bf16)
    if c in ("UZ_OP_NEAREST_FWD", "UZ_OP_NEAREST_BWD"):
        C, N, H, W, f = i[0], i[3], i[4], i[5], i[6]
        return f4 * C * N * H * W * (1 + f * f)
    if c in ("UZ_OP_AVGPOOL3D_FWD", "UZ_OP_AVGPOOL3D_BWD"):
        C, D, H, W = i[0], i[3], i[4], i[5]
        hi, lo = (0, 1) if c == "UZ_OP_AVGPOOL3D_FWD" else (1, 0)      # format bits: (x, y) / (dy, dx)
        return C * D * H * W * (_esz(op, hi) + 0.125 * _esz(op, lo))
    if c in ("UZ_OP_DEPTH_LERP_FWD", "UZ_OP_DEPTH_LERP_BWD"):
        C, D, H, W = i[0], i[3], i[4], i[5]
        lo, hi = (0, 1) if c == "UZ_OP_DEPTH_LERP_FWD" else (1, 0)     # D slices on the low side, 2 D on the high side
        return C * D * H * W * (_esz(op, lo) + 2 * _esz(op, hi))
    if c in ("UZ_OP_NEAREST3D_FWD", "UZ_OP_NEAREST3D_BWD"):
        C, D, H, W, f, fz = i[0], i[3], i[4], i[5], i[6], i[7]
        return f4 * C * D * H * W * (1 + f * f * fz)
    if c == "UZ_OP_ADD_VIEWS":
        C, N, H, W = i[3], i[4], i[5], i[6]
        return f4 * C * N * H * W * (3 if i[1] else 2)
    return 0.0


FAMILY = {"UZ_OP_CONV_FWD": "conv_fwd", "UZ_OP_CONV_BWD_DATA": "conv_dgrad", "UZ_OP_CONV_BWD_WEIGHT": "conv_wgrad",
          "UZ_OP_BN_RELU_FWD": "bn_relu_fwd", "UZ_OP_BN_RELU_BWD": "bn_relu_bwd", "UZ_OP_RELU_BWD": "relu_bwd",
          "UZ_OP_AVGPOOL_FWD": "resample", "UZ_OP_AVGPOOL_BWD": "resample", "UZ_OP_BILINEAR_FWD": "resample",
          "UZ_OP_BILINEAR_BWD": "resample", "UZ_OP_NEAREST_FWD": "resample", "UZ_OP_NEAREST_BWD": "resample",
          "UZ_OP_AVGPOOL3D_FWD": "resample", "UZ_OP_AVGPOOL3D_BWD": "resample", "UZ_OP_DEPTH_LERP_FWD": "resample",
          "UZ_OP_DEPTH_LERP_BWD": "resample", "UZ_OP_NEAREST3D_FWD": "resample", "UZ_OP_NEAREST3D_BWD": "resample",
          "UZ_OP_ADD_VIEWS": "add_copy"}


def binding_roof(plan, L):
    """Effective matrix-pipe roof of the step: sum(flops) / sum(flops_i / roof_i) over the conv ops of the fwd + bwd tapes,
    plus the FLOP share that runs on each pipe."""
    tot = t_at_roof = 0.0
    share = {"fp32_mfma": 0.0, "split_fp16_mfma": 0.0, "valu_streaming_heads": 0.0}
    for ops in (plan.fwd_ops, plan.loss_ops, plan.bwd_ops):
        for o in ops:
            fl = conv_flops(o)
            if not fl:
                continue
            roof = conv_roof(o, L)
            if roof is None:
                share["valu_streaming_heads"] += fl
                continue                                     # memory-bound heads: no matrix-pipe roof, < 0.1 % of the FLOPs
            share["fp32_mfma" if roof == PEAK_F32_MFMA_TFLOPS else "split_fp16_mfma"] += fl
            tot += fl
            t_at_roof += fl / roof
    allf = sum(share.values()) or 1.0
    return tot / t_at_roof, {k: round(v / allf, 4) for k, v in share.items()}


def cu_share(op, L):
    """Share of the chip's 256 CUs the launch of a tape op can occupy: 1 for everything except the split-path weight gradient, whose grid
    is cut into (channel tiles) x (the slab count the op carries) workgroups of one per CU (uz_set_wgrad_target; PHiSeg: 128 on purpose)."""
    if op["code"] != "UZ_OP_CONV_BWD_WEIGHT":
        return 1.0
    try:
        kind, cin, cout, n, h, w, ks = conv_dims(op)
        if ks != 3 or L.uz_conv_route(2, cin, cout, n, h, w, ks) != 1:
            return 1.0
        ct = 32 if (cin <= 32 or cout <= 32) else 64
        grid = -(-cout // ct) * -(-cin // ct) * int(L.uz_conv_bwd_weight_slabs(cin, cout, n, h, w, ks))
        return min(1.0, grid / 256.0) if grid > 0 else 1.0
    except Exception:
        return 1.0


def profile_families(net, plan, L, reps=3, burst=4):
    """Live per-family timing: replay the fwd / bwd tapes one op at a time between two HIP events on the launch stream
    (torch.cuda.Event records on torch's current stream, which IS the stream the tape is launched on).
    Two figures per op: `ms` = best of `reps` SINGLE launches - what the HBM fractions are computed from; it carries the
    ~5 us an event pair and a launch onto an idle queue cost, so it under-states short kernels - and `ms_burst` = best burst of
    `burst` identical back-to-back launches / burst: launch overhead amortised, but launches 2..burst re-read what the first left in
    the 256 MB memory-side cache, so it is CACHE-WARM and over-states bandwidth; reported beside the first, labelled, never as the
    HBM fraction."""
    import ctypes as C
    import torch
    from unet_zoo_amd import _ffi
    stream = C.c_void_p(net._stream())
    fam, heaviest, heaviest_full = {}, None, None

    def timed(tape, n_launch):
        best = None
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _ffi.check(L.uz_run_tape(tape, n_launch, stream), "profile op")
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / n_launch
            best = ms if best is None else min(best, ms)
        return best
    for which, ops in (("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops)):
        arr, n = plan.tapes[which]
        for k in range(n):
            name = FAMILY.get(ops[k]["code"], "other")
            best = timed((type(arr[0]) * 1)(arr[k]), 1)
            d = fam.setdefault(name, dict(ms=0.0, flops=0.0, t_roof=0.0, bytes=0.0, launches=0, ms_large=0.0, bytes_large=0.0, n_large=0, ms_large_burst=0.0, chip_ms=0.0))
            fl = conv_flops(ops[k])
            d["ms"] += best
            d["chip_ms"] += best * cu_share(ops[k], L)
            d["bytes"] += op_bytes(ops[k])
            d["launches"] += 1
            if op_bytes(ops[k]) >= LARGE_OP_BYTES:            # streaming ops big enough to be bandwidth- rather than latency-bound
                d["ms_large"] += best
                d["bytes_large"] += op_bytes(ops[k])
                d["n_large"] += 1
                if not fl and ops[k]["code"] != "UZ_OP_EVENT_RECORD":
                    d["ms_large_burst"] += timed((type(arr[0]) * burst)(*([arr[k]] * burst)), burst)
            if fl:
                roof = conv_roof(ops[k], L)
                if roof is not None:
                    d["flops"] += fl
                    d["t_roof"] += fl / (roof * 1e12)                 # seconds at the roof
                    if heaviest is None or best > heaviest[0]:
                        heaviest = (best, which, k, fl, roof)
                    # the longest launch among those whose grid covers the whole chip: a split-path weight gradient is cut into
                    # uz_get_wgrad_target() workgroups (PHiSeg: 128 of 256 CUs, on purpose - the rest of the chip stays with the other lanes)
                    reduced = ops[k]["code"] == "UZ_OP_CONV_BWD_WEIGHT" and roof != PEAK_F32_MFMA_TFLOPS and L.uz_get_wgrad_target() < 256
                    if not reduced and (heaviest_full is None or best > heaviest_full[0]):
                        heaviest_full = (best, which, k, fl, roof)
    # The DOMINANT launch is the one that takes the most of the CHIP's time: duration x the share of the CUs its grid occupies.  A weight
    # gradient cut into 128 workgroups runs 1.25 ms on HALF the CUs (0.63 chip-ms) while other lanes use the rest; the data gradient of the
    # same layer holds all 256 CUs for 0.81 ms.  The longest launch is reported beside it (roofline.longest_launch_reduced_grid).
    if heaviest is not None and heaviest_full is not None and heaviest[1:3] != heaviest_full[1:3]:
        share = min(1.0, L.uz_get_wgrad_target() / 256.0)
        if heaviest_full[0] >= heaviest[0] * share:
            heaviest, heaviest_full = heaviest_full, heaviest
    return fam, heaviest, heaviest_full


def dominant_kernel_live(net, plan, L, heaviest, reps=20):
    """The single heaviest kernel launch of the step (longest conv op of the tapes), re-timed live: `reps` back-to-back
    launches of that tape op between two HIP events on its launch stream.  achieved = algorithmic FLOPs per launch /
    average launch duration; peak = the roof of the pipe the op runs on.  `traffic` = PMC-measured HBM bytes per launch
    when a committed PMC pass covers this kernel (profiles/*pmc_traffic.json), else null."""
    import ctypes as C
    import torch
    from unet_zoo_amd import _ffi
    _, which, k, flops, roof = heaviest
    ops = plan.fwd_ops if which == "fwd" else plan.bwd_ops
    arr, _n = plan.tapes[which]
    one = (type(arr[0]) * 1)(arr[k])
    stream = C.c_void_p(net._stream())
    # the ops of the same scheduling group in front of it, once: a backward convolution reads its dy from the lane's scratch,
    # which after a full step holds another layer's tensor (tools/layer_profile.py does the same)
    j = k
    while j > 0 and ops[j - 1]["gid"] == ops[k]["gid"]:
        j -= 1
    if j < k:
        _ffi.check(L.uz_run_tape((type(arr[0]) * (k - j))(*[arr[i] for i in range(j, k)]), k - j, stream), "dominant op: group prefix")
    _ffi.check(L.uz_run_tape(one, 1, stream), "dominant op")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _ffi.check(L.uz_run_tape(one, 1, stream), "dominant op")
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    kind, cin, cout, n, h, w, ks = conv_dims(ops[k])
    split = roof != PEAK_F32_MFMA_TFLOPS
    kname = {0: "conv_split_kernel (+ pack_weights_kernel)" if split else "conv_mfma_kernel",
             1: "conv_split_kernel, data gradient (+ pack_weights_kernel)" if split else "conv_mfma_kernel, data gradient",
             2: "wgrad_split_kernel (+ wgrad_reduce)" if split else "wgrad_fast_kernel (+ wgrad_reduce)"}[kind]
    alg_bytes = conv_bytes(ops[k])
    out = dict(kernel=kname, op={0: "forward", 1: "data gradient", 2: "weight gradient"}[kind],
               layer=f"{ks}x{ks} {cin}->{cout} @ {n}x{h}x{w}", flops_per_launch=flops, avg_launch_ms=round(ms, 4),
               achieved=round(flops / ms / 1e9, 2), peak=round(roof, 1), unit="TFLOP/s", frac=round(flops / ms / 1e9 / roof, 4),
               algorithmic_bytes=alg_bytes, traffic=None,
               peak_note=("dense bf16 MFMA peak 2500 TFLOP/s, one product per MAC (UZ_CONV_MATH=bf16)" if roof == PEAK_F16_MFMA_TFLOPS else
                          "dense fp16 MFMA peak 2500 TFLOP/s / 3 piece products per fp32 product" if split else "fp32 MFMA peak"))
    if split:
        out["fp16_mfma_tflops"] = round((1 if roof == PEAK_F16_MFMA_TFLOPS else SPLIT_PRODUCTS) * flops / ms / 1e9, 1)
        if roof != PEAK_F16_MFMA_TFLOPS:
            # informational, not the contract's peak: what the chip sustains with its matrix pipe saturated on operand pieces like these
            # (tools/micro/pingpong.hip on one MI355X: 1.51 - 1.60 PF/s of fp16 MFMAs at 1.5 - 1.7 GHz, profiles/r4_call107_pingpong.txt)
            out["sustained_ceiling_note"] = ("measured power-limited ceiling of the fp16 matrix pipe on realistic split pieces: 1510 - 1600 TFLOP/s = "
                                             "503 - 533 TFLOP/s fp32-equivalent = 0.60 - 0.64 of `peak` (tools/micro/pingpong.hip)")
    if roof == PEAK_F16_MFMA_TFLOPS:
        # one product per MAC puts these layers on the memory side of the ridge for narrow channel counts: report the HBM view too
        out["hbm_view"] = dict(achieved_gbs=round(alg_bytes / ms / 1e6, 1), peak_gbs=HBM_PEAK_GBS, frac=round(alg_bytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                               note="algorithmic bytes (input + output tensor once, at their storage width) / launch duration against 8 TB/s")
    # bf16-STORAGE volume path: the PMC passes of tools/prof_b16.sh on the heaviest PHiSeg3D layer (profiles/r4_pmc_traffic_b16.json)
    op_i = ops[k]["i"]
    if len(op_i) > 13 and op_i[13]:
        try:
            pname = next(n for n in ("r6_pmc_traffic_b16.json", "r5_pmc_traffic_b16.json", "r4_pmc_traffic_b16.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pname)))
            if f"{cin} -> {cout}" in pmc.get("layer", "") and kind in (0, 1, 2):
                key = "wgrad_split_kernel<32, 64, 1, 2, 2>" if kind == 2 else next(k_ for k_ in ("conv_b16_db_kernel_4_512_32", "conv_b16_db_kernel_2_512_32", "conv_b16_kernel_2_512_32") if k_ in pmc["kernels"])
                e = pmc["kernels"][key]
                out.update(traffic=e["hbm_bytes"], traffic_source="profiles/" + pname, kernel=key,
                           traffic_note="FETCH_SIZE + WRITE_SIZE per launch (L2 fills: Infinity-Cache hits included), calibrated; x %.2f of the algorithmic bytes - "
                                        "the depth window reads every input slice for three output slices" % e["ratio"])
                return out
        except Exception:
            pass
    for tabname in ("r6_layer_table.json", "r5_layer_table.json", "r4_layer_table.json", "r3_layer_table.json"):      # per-layer dispatch table (tools/prof_layers.sh), keyed on layer and direction; newest first
        try:
            tab = json.load(open(os.path.join(ROOT, "profiles", tabname)))
            for row in tab["layers"]:
                if (row["kind"], row["cin"], row["cout"], row["n"], row["h"], row["w"], row["ks"]) == (kind, cin, cout, n, h, w, ks):
                    # (a COMMITTED table, not a measurement of this run: its git blob hash says which one, `profiled.kernel` which kernel it
                    #  timed - regenerate it in the evidence call behind the last kernel commit)
                    raw = open(os.path.join(ROOT, "profiles", tabname), "rb").read()
                    import hashlib
                    blob = hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest()
                    out.update(traffic=row["hbm_bytes"], traffic_source="profiles/" + tabname, traffic_table_git_blob=blob, kernel=row["dominant_kernel"],
                               profiled=dict(avg_launch_us=row["avg_launch_us"], mfma_busy_fraction=row["mfma_busy_fraction"],
                                             traffic_over_algorithmic=row["traffic_over_algorithmic"]))
                    return out
        except Exception:
            pass
    for prof in ("r2_pmc_traffic.json", "r1_pmc_traffic.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", prof)))
            if f"{cin} -> {cout}" not in pmc.get("layer", "") or f"N={n}" not in pmc.get("layer", ""):
                continue
            key = {0: "conv_split_kernel<2> f16 (forward)", 1: "conv_split_kernel<2> f16 (forward)", 2: "wgrad_split_kernel<32,64> f16 (weight gradient, slabs)"}[kind] \
                if split else {0: "conv_mfma_kernel<3,2,2,false> (forward)", 1: "conv_mfma_kernel<3,2,2,true> (data gradient)",
                               2: "wgrad_fast_kernel<2,2,4,32> (weight gradient, slabs)"}[kind]
            out.update(traffic=pmc["kernels"][key]["hbm_bytes"], traffic_source="profiles/" + prof)
            break
        except Exception:
            pass
    return out


def conv_math():
    return os.environ.get("UZ_CONV_MATH", "default")


def fp32_only_leg(args):
    """Same benchmark in a child process with UZ_CONV_MATH=f32 (every convolution on the fp32 MFMA kernels).  Started
    BEFORE this process touches the GPU."""
    env = dict(os.environ, UZ_CONV_MATH="f32")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--batch", str(args.batch), "--model", args.model, "--skip-cpu", "--no-profile", "--no-f32-leg"]
    if args.allow_experiment:
        cmd.append("--allow-experiment")
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return dict(value=d["value"], unit=d["unit"], ms_per_step=d["ms_per_step"], frac=d["roofline"]["frac"],
                    note="identical run with UZ_CONV_MATH=f32: all convolutions on v_mfma_f32_32x32x2_f32; frac is against the 157.3 TFLOP/s fp32-MFMA peak")
    except Exception as e:                                   # never fail the headline line because of the extra leg
        return dict(error=str(e)[:200])


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU, torchrun's environment
    contract) and relay rank 0's JSON line.  Runs BEFORE this process makes any GPU call - the children are new processes,
    never a re-exec of a GPU-initialised one.

    Every rank is watched: one that dies (RCCL init, out of memory) leaves its peers waiting in a collective forever, and one
    that hangs INSIDE ncclCommInitRank / a collective never exits at all.  So (a) a non-zero exit stops the survivors at once,
    (b) a wall-clock deadline (UZ_BENCH_DEADLINE_S, default 900 s; the driver's own default run finishes in well under a minute
    per GPU count) stops everything when rank 0 has not finished by then - in both cases the command exits non-zero with every
    rank's exit code and the tail of its stderr in the message instead of hanging until the caller's timeout."""
    import socket
    import tempfile
    import threading
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    deadline_s = float(os.environ.get("UZ_BENCH_DEADLINE_S", "900"))
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), UZ_BENCH_SELF_LAUNCHED="1")
        ef = tempfile.TemporaryFile(mode="w+", prefix=f"uz_bench_rank{r}_")      # every rank's stderr is kept (a file: no pipe to fill up)
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=ef, text=True))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed, timed_out, t0 = None, False, time.monotonic()
    while failed is None and not timed_out and any(q.poll() is None for q in procs):
        for r, q in enumerate(procs):
            if q.poll() not in (None, 0):
                failed = r
        timed_out = time.monotonic() - t0 > deadline_s
        time.sleep(0.2)
    if failed is not None or timed_out:
        for q in procs:
            if q.poll() is None:
                q.kill()
    rcs = [q.wait() for q in procs]
    reader.join(timeout=10)
    sys.stdout.write("".join(c for c in chunks if c))
    sys.stdout.flush()

    def tail(f, lines=12):
        f.seek(0)
        return "".join(f.readlines()[-lines:])
    if failed is not None or timed_out:
        what = (f"rank {failed} exited with code {rcs[failed]}; the other ranks were stopped" if failed is not None
                else f"no result after {deadline_s:.0f} s (UZ_BENCH_DEADLINE_S): every rank was stopped")
        print(f"bench.py: {what}", file=sys.stderr)
        for r, f in enumerate(errs):
            t = tail(f)
            print(f"---- rank {r}: exit code {rcs[r]}, stderr tail:\n{t if t else '(empty)'}", file=sys.stderr)
    else:
        sys.stderr.write(tail(errs[0], 50))                  # rank 0's own diagnostics (warnings) are passed through
    for f in errs:
        f.close()
    return 1 if timed_out else max(abs(rc) for rc in rcs)


def build_info():
    """uz_build_info() of the library this process loads: what the binary is (variant, piece products per MAC, experiment code, source hash)."""
    import ctypes as C
    from unet_zoo_amd import _ffi
    buf = C.create_string_buffer(1024)
    _ffi.lib().uz_build_info(buf, 1024)
    return json.loads(buf.value.decode())


def uz_env():
    """Every UZ_* / queue variable set in this process's environment (the line's config.env): a reader sees which knobs shaped the run."""
    keys = sorted(k for k in os.environ if k.startswith("UZ_") or k in ("GPU_MAX_HW_QUEUES", "DEBUG_HIP_FORCE_GRAPH_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY"))
    return {k: os.environ[k] for k in keys}


def product_guard(allow):
    """The line must prove it ran the product: a run with a work-skipping diagnostic (UZ_DIAG_SKIP), another library (UZ_LIB) or a library
    built with timing-only / diagnostic code (uz_build_info().experiment) is refused - exit status 2 - unless --allow-experiment is given,
    and then the line carries "experiment": true.  Returns the reasons (empty list: a product run)."""
    reasons = [f"{k} is set" for k in ("UZ_DIAG_SKIP", "UZ_LIB") if os.environ.get(k)]
    try:
        info = build_info()
        if info.get("experiment"):
            reasons.append("the library is an experiment build: " + json.dumps({k: v for k, v in info.items() if k != "source_hash"}))
    except Exception as e:                                   # a library without uz_build_info is not the product either
        reasons.append(f"uz_build_info failed: {e}")
    if reasons and not allow:
        print("bench.py: refusing to report a benchmark line - " + "; ".join(reasons) + " (pass --allow-experiment to run anyway; the line is then marked)",
              file=sys.stderr)
        raise SystemExit(2)
    return reasons


def dry_run(args, rank, world, global_batch, experiment=()):
    """UZ_BENCH_DRY=1 (CPU test hook, never used by the driver): the launcher / rendezvous / timing / reporting skeleton of main()
    with a sleep in place of the training step - checks that `--gpus N` really runs N ranks and reports that number."""
    import torch
    import torch.distributed as dist
    if os.environ.get("UZ_BENCH_DRY_FAIL_RANK") == str(rank):      # test hook: a rank that dies before the rendezvous
        print(f"dry run: rank {rank} fails on purpose", file=sys.stderr)
        sys.exit(3)
    if os.environ.get("UZ_BENCH_DRY_HANG_RANK") == str(rank):      # test hook: a rank stuck before the rendezvous (as inside ncclCommInitRank)
        print(f"dry run: rank {rank} hangs on purpose", file=sys.stderr, flush=True)
        time.sleep(3600)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    elapsed = time.perf_counter() - t0
    nranks = 1
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        nranks = dist.get_world_size()
    if rank == 0:
        print(json.dumps(dict(metric="dry run (no GPU work)", value=round(args.batch * world * args.steps / elapsed, 2), unit="images/s", n_gpus=world,
                              steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * elapsed / args.steps, 3), higher_is_better=True,
                              scaling="strong" if args.strong else "weak", vs_baseline=None, dtype="none", data="none",
                              config=dict(workload="dry run", batch_per_gpu=args.batch, global_batch=global_batch, parallelism=f"dp{world}", build=build_info(), env=uz_env()),
                              dp=dict(backend="gloo", nranks=nranks, launcher="self" if os.environ.get("UZ_BENCH_SELF_LAUNCHED") else "external"),
                              **({"experiment": True} if experiment else {}))))
    if world > 1:
        dist.destroy_process_group()


def usable_cores():
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(model, batch, budget_s=30.0):
    """CPU oracle timed on the host cores on a bounded sample (about `budget_s` seconds of CPU work).
    Only this leg of bench.py imports oracle/."""
    import torch
    import oracle
    from unet_zoo_amd.models.phiseg import phiseg_spec
    from unet_zoo_amd.models.unet import unet_spec
    from unet_zoo_amd.models.probabilistic_unet import probunet_spec
    cores = usable_cores()
    threads = min(cores, 32)                  # oneDNN / OpenMP stop scaling (and thrash) far below 256 threads
    torch.set_num_threads(threads)
    if model == "phiseg3d":
        return cpu_baseline_3d(threads, cores, budget_s)
    spec = {"phiseg": lambda: phiseg_spec(1, 2, FILTERS7), "unet": lambda: unet_spec(1, 2, FILTERS4),
            "probunet": lambda: probunet_spec(1, 2, FILTERS7, 6, 3)}[model]()
    sd = oracle.deterministic_state_dict(spec, seed=3)
    leaves = {}
    for k, v in sd.items():
        t = v.clone()
        if t.dtype.is_floating_point and "running_" not in k:
            t.requires_grad_(True)
        leaves[k] = t
    shapes = oracle.phiseg_eps_shapes(batch, 128, 128)
    state, times = {}, []
    t_start = time.perf_counter()
    for step in range(4):
        eshapes = {"phiseg": shapes + shapes, "unet": None, "probunet": [(batch, 6)]}[model]
        x, mask, eps = oracle.synthetic_batch(batch, 128, 128, seed=100 + step, eps_shapes=eshapes)
        xt, mt = torch.from_numpy(x), torch.from_numpy(mask)
        t0 = time.perf_counter()
        if model == "phiseg":
            e = [torch.from_numpy(a) for a in eps]
            out = oracle.phiseg_forward(leaves, xt, mt, dict(posterior=e[:5], prior=e[5:]))
            total, _ = oracle.phiseg_loss(out, mt)
        elif model == "unet":
            total = oracle.unet_loss(oracle.unet_forward(leaves, xt), mt)
        else:
            out = oracle.probunet_forward(leaves, xt, mt, bn_train=True)
            total, _ = oracle.probunet_loss(leaves, out, mt, torch.from_numpy(eps[0]), bn_train=True)
        for v in leaves.values():
            v.grad = None
        total.backward()
        params = {k: v for k, v in leaves.items() if v.requires_grad}
        new = oracle.adam_reference_step(params, {k: v.grad for k, v in params.items()}, state)
        for k, v in new.items():
            leaves[k] = v.requires_grad_(True)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:
            break
    timed = times[1:] if len(times) > 1 else times           # drop the warm-up step when there was time for more
    sec = sum(timed) / len(timed)
    return dict(value=round(batch / sec, 3), unit="images/s", cores=threads, kind="port",
                sample=f"CPU oracle (functional torch fp32 restatement of the reference graph), {model} 128x128 batch {batch}, "
                       f"{len(timed)} timed step(s){' after 1 warm-up' if len(times) > 1 else ' (warm-up only: budget exhausted)'}, "
                       f"fwd+loss+bwd+Adam, {threads} threads of {cores} usable cores, {sec:.2f} s/step")


def cpu_baseline_3d(threads, cores, budget_s):
    """PHISeg3D on the host: the CPU oracle (oracle/refgraph3d.py) on ONE volume of the benchmark's architecture.  A full
    128x128x64 volume costs ~13 TFLOP per step - minutes on host cores - so the bounded sample is a 64x64x32 volume (1/8 of the
    voxels, identical network) and the value is scaled to full-volume equivalents by the voxel ratio."""
    import torch
    import oracle
    from oracle import refgraph3d as R3
    from unet_zoo_amd.models.phiseg3D import phiseg3d_spec
    sd = oracle.deterministic_state_dict(phiseg3d_spec(4, 3, FILTERS3D, 5), seed=3)
    leaves = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running_" not in k else v.clone()) for k, v in sd.items()}
    dhw = tuple(n // 2 for n in DHW3D)
    shapes = R3.phiseg3d_eps_shapes(*dhw, 5, 5)
    x, onehot, lab, eps = R3.synthetic_volume(4, 3, dhw, 100, shapes + shapes)
    times, t_start = [], time.perf_counter()
    for step in range(3):
        t0 = time.perf_counter()
        out = R3.phiseg3d_forward(leaves, torch.from_numpy(x), torch.from_numpy(onehot), [torch.from_numpy(e) for e in eps])
        total, _ = R3.phiseg3d_loss(out, torch.from_numpy(lab), num_classes=3)
        for v in leaves.values():
            v.grad = None
        total.backward()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:
            break
    timed = times[1:] if len(times) > 1 else times
    sec = sum(timed) / len(timed)
    return dict(value=round(0.125 / sec, 4), unit="volumes/s", cores=threads, kind="port",
                sample=f"CPU oracle (functional torch fp32 restatement of models/phiseg3D.py), same network on a 64x64x32 volume (1/8 of the "
                       f"voxels; value = 1/8 volume per {sec:.2f} s, i.e. scaled to 128x128x64 equivalents), {len(timed)} timed step(s)"
                       f"{' after 1 warm-up' if len(times) > 1 else ''}, fwd+loss+bwd (no Adam), {threads} threads of {cores} usable cores")


def build(model, reversible=False):
    if model == "phiseg3d":
        from unet_zoo_amd.models.phiseg3D import PHISeg3D
        return PHISeg3D(4, 3, FILTERS3D, latent_levels=5, image_size=(4, *DHW3D), reversible=reversible)
    from unet_zoo_amd.models.phiseg import PHISeg
    from unet_zoo_amd.models.unet import Unet
    from unet_zoo_amd.models.probabilistic_unet import ProbabilisticUnet
    if model == "phiseg":
        return PHISeg(input_channels=1, num_classes=2, num_filters=FILTERS7, latent_levels=5, image_size=(1, 128, 128))
    if model == "unet":
        return Unet(1, 2, FILTERS4)
    return ProbabilisticUnet(1, 2, FILTERS7, latent_dim=6, no_convs_fcomb=3, image_size=(1, 128, 128))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (weak scaling); default 32 (phiseg3d: 1 volume)")
    ap.add_argument("--reversible", action="store_true", help="phiseg3d: reversible blocks (the reference's BraTS experiment sets use_reversible)")
    ap.add_argument("--model", choices=sorted(MODELS), default="phiseg")
    ap.add_argument("--tune-schedule", type=int, default=int(os.environ.get("UZ_TUNE_SCHEDULE", "8")),
                    help="rounds of profile-guided lane scheduling before the timed region (Engine.tune_schedule; 0 = the static cost model)")
    ap.add_argument("--no-graphs", action="store_true", help="launch kernels eagerly instead of hipGraph replay")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-family HIP-event pass")
    ap.add_argument("--skip-cpu", action="store_true", help="skip the CPU-oracle baseline leg")
    ap.add_argument("--cpu-batch", type=int, default=32, help="batch of the CPU-oracle leg (BASELINE.md section 3: 32)")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the extra fp32-MFMA-only measurement (child process)")
    ap.add_argument("--no-overlap", action="store_true", help="data parallel: one blocking all-reduce after backward instead of bucketed overlap")
    ap.add_argument("--conv-math", choices=["default", "f32", "split", "bf16"], default=None,
                    help="arithmetic of the large 3x3 convolutions (= UZ_CONV_MATH): default = fp32-accurate fp16 split; bf16 = one bf16 piece per "
                         "operand, fp32 accumulation.  phiseg3d defaults to bf16 (BASELINE configs[4] is quoted in bf16), everything else to default")
    ap.add_argument("--storage", choices=["f32", "bf16"], default=None,
                    help="phiseg3d with --conv-math bf16: keep the large volume tensors (activations, their gradients, dy) in bf16 STORAGE "
                         "(= UZ_STORE_B16=1; BASELINE configs[4] 'bf16': the default of --model phiseg3d), or everything in fp32")
    ap.add_argument("--strong", action="store_true", help="strong scaling (SURVEY 8d): the GLOBAL batch stays at --batch (default 32), "
                                                          "each of the N GPUs takes batch / N images")
    ap.add_argument("--allow-experiment", action="store_true",
                    help="report a line although a work-skipping / timing-only knob is in play (UZ_DIAG_SKIP, UZ_LIB, a library built with "
                         "UZ_EXP_* / UZ_DIAG): the line is then marked \"experiment\": true.  Without this flag such a run exits with status 2.")
    args = ap.parse_args()
    experiment = product_guard(args.allow_experiment)

    if args.conv_math is None and args.model == "phiseg3d" and "UZ_CONV_MATH" not in os.environ:
        args.conv_math = "bf16"
    if args.conv_math and args.conv_math != "default":
        os.environ["UZ_CONV_MATH"] = args.conv_math          # read by the library when it first routes a convolution; inherited by the ranks
    if args.storage is None and args.model == "phiseg3d" and os.environ.get("UZ_CONV_MATH") == "bf16" and "UZ_STORE_B16" not in os.environ:
        args.storage = "bf16"
    if args.storage:
        os.environ["UZ_STORE_B16"] = "1" if args.storage == "bf16" else "0"          # read when a plan is built; inherited by the ranks
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would report a GPU count that is not the number of ranks")
    M = dict(MODELS[args.model])
    vol = args.model == "phiseg3d"
    if args.batch is None:
        args.batch = 1 if vol else 32
    if vol and args.batch != 1:
        raise SystemExit("phiseg3d runs one volume per GPU and step (BASELINE configs[4]: batch 8 on 8 GPUs)")
    global_batch = args.batch * world
    if args.strong:
        if vol or args.batch % world:
            raise SystemExit(f"--strong: the global batch {args.batch} must divide over {world} GPUs (2-D models only)")
        global_batch, args.batch = args.batch, args.batch // world

    if os.environ.get("UZ_BENCH_DRY") == "1":
        return dry_run(args, rank, world, global_batch, experiment)

    # the fp32-MFMA-only comparison leg runs in a child process BEFORE this process initialises the GPU
    f32_leg = None
    if world == 1 and not args.no_f32_leg and conv_math() != "f32" and not vol:
        f32_leg = fp32_only_leg(args)

    import torch
    # Test hooks (not used by the driver): UZ_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and UZ_BENCH_BACKEND=gloo
    # replaces RCCL, so that the multi-rank code path can be exercised on a one-GPU box.
    if os.environ.get("UZ_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("UZ_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        # Control plane = a gloo group on the host (unique-id hand-off, barriers, the max over ranks of the elapsed time).
        # The ONLY RCCL communicator of the process is the one GradSync opens through the C ABI (uz_comm_init); parameters are
        # broadcast and gradients averaged on it.  UZ_BENCH_BACKEND=gloo (test hook) moves the data plane to gloo as well.
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    if world > 1:
        # data parallel: train on a created stream - the overlapped gradient exchange needs a second stream, and the legacy
        # default stream serialises against it (dp.GradSync.sync)
        torch.cuda.set_stream(torch.cuda.Stream(torch.device("cuda", local_rank)))
    from unet_zoo_amd import _ffi
    from unet_zoo_amd.synthetic import synthetic_batch
    from unet_zoo_amd.optim import FusedAdam
    L = _ffi.lib()

    torch.manual_seed(1234)          # same initial weights on every rank (DP replicas)
    net = build(args.model, args.reversible)
    net.train()
    if world > 1:
        net.set_data_parallel(True, overlap=not args.no_overlap, backend="rccl" if backend == "nccl" else "torch")
        net._dp.broadcast_params()
        if net._dp.backend == "rccl" and net._dp.nranks() != world:
            raise SystemExit(f"RCCL reports {net._dp.nranks()} ranks but WORLD_SIZE={world}")
    if not args.no_graphs:
        net.enable_graphs(True)
    opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)        # train_model.py:49
    dev = torch.device("cuda", local_rank)
    if vol:
        from unet_zoo_amd.synthetic import synthetic_volume
        x, onehot, mask = (torch.from_numpy(a).to(dev) for a in synthetic_volume(4, 3, DHW3D, seed=20201005 + rank))
    else:
        x, mask, _ = synthetic_batch(args.batch, 128, 128, seed=20201004 + rank)
        x, mask = torch.from_numpy(x).to(dev), torch.from_numpy(mask).to(dev)

    def step():
        if vol:
            net.forward(x, onehot, training=True)
        elif args.model == "unet":
            net.forward(x)
        else:
            net.forward(x, mask, training=True)
        loss = net.loss(mask)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(max(args.warmup, 3 if not args.no_graphs else 1)):
        loss = step()
    tuned = None
    if args.tune_schedule > 0 and not args.no_graphs and net.replay_mode == "lanes":
        # profile-guided lane schedule (Engine.tune_schedule): set-up work like the graph capture, untimed; under data parallelism the call is
        # COLLECTIVE - every measured duration is max-reduced over the ranks (dp.max_over_ranks), so all ranks end with one schedule and
        # one bucket exchange order
        tuned = net.tune_schedule(step, rounds=args.tune_schedule)
        for _ in range(2):
            loss = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss.detach())

    extra = {}
    train_plan = net._cur
    if args.model == "probunet" and rank == 0:
        # "8 posterior samples" (BASELINE configs[2]): after one forward, 8 x [z = posterior.rsample(); fcomb(features, z)]
        # = reconstruct(calculate_posterior=True) (probabilistic_unet.py:272-283), timed separately from the train step
        net.eval()
        with torch.no_grad():
            net.forward(x, mask, training=False)
            for _ in range(3):
                net.reconstruct(calculate_posterior=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                for _ in range(8):
                    net.reconstruct(calculate_posterior=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / reps
        extra["decode_8_posterior_samples"] = dict(ms=round(1e3 * dt, 3), decoded_images_per_s=round(8 * args.batch / dt, 1),
                                                   note="8 x reconstruct(calculate_posterior=True) on the cached U-Net features, eval mode, batch %d" % args.batch)
        net.train()

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        ips = args.batch * world * args.steps / elapsed
        per_gpu = ips / world
        plan = train_plan
        if vol:                                              # work per volume from the plan's own ops
            M["gflop"] = sum(conv_flops(o) for ops in (plan.fwd_ops, plan.bwd_ops) for o in ops) / 1e9
            M["gb_img"] = sum(op_bytes(o) + conv_bytes(o) for ops in (plan.fwd_ops, plan.bwd_ops) for o in ops) / 1e9
            M["gb_step"] = 28.0 * net._ptab.n_params / 1e9          # Adam: p, g, m, v read + p, m, v written
        achieved = per_gpu * M["gflop"] / 1e3
        eff_peak, share = binding_roof(plan, L)
        roof = dict(bound="mfma", achieved=round(achieved, 3), peak=round(eff_peak, 1), unit="TFLOP/s",
                    frac=round(achieved / eff_peak, 4), traffic=None,
                    frac_vs_fp32_mfma=round(achieved / PEAK_F32_MFMA_TFLOPS, 4), flop_share_by_pipe=share,
                    hbm_fraction=round((per_gpu * M["gb_img"] + (per_gpu / args.batch) * M["gb_step"]) / HBM_PEAK_GBS, 4),
                    note=f"step-level, per GPU: achieved = images/s x {M['gflop']:.2f} GFLOP/image (fp32-equivalent); peak = binding roof of the "
                         "conv ops as routed: 157.3 TFLOP/s (fp32 MFMA) or 2500/3 = 833.3 TFLOP/s (split-fp16, 3 fp16 products per fp32 "
                         "product), weighted by FLOPs (harmonic); frac_vs_fp32_mfma = achieved / 157.3 as BASELINE.md section 2 defines it; "
                         "hbm_fraction uses the unfused-graph bytes and cannot exceed ~0.21 in fp32")
        if not args.no_profile and world == 1:
            try:
                fam, heaviest, heaviest_full = profile_families(net, plan, L)
                fams = {}
                for k, d in fam.items():
                    e = dict(ms_per_step=round(d["ms"], 3), launches=d["launches"])
                    if d["flops"]:
                        e["tflops"] = round(d["flops"] / d["ms"] / 1e9, 2)
                        e["binding_roof_tflops"] = round(d["flops"] / d["t_roof"] / 1e12, 1)
                        e["frac_of_binding_roof"] = round(d["t_roof"] * 1e3 / d["ms"], 4)
                        if d["chip_ms"] < 0.98 * d["ms"]:
                            e["chip_ms_per_step"] = round(d["chip_ms"], 3)
                            e["frac_on_occupied_cus"] = round(d["t_roof"] * 1e3 / d["chip_ms"], 4)
                            e["grid_note"] = ("ms_per_step sums isolated launch durations; the split-path launches of this family are cut into fewer workgroups than CUs "
                                              "on purpose (uz_set_wgrad_target) and leave the rest of the chip to the other lanes: chip_ms_per_step = sum of duration x "
                                              "share of the 256 CUs the grid can occupy, frac_on_occupied_cus = time at the roof / chip_ms_per_step")
                    elif d["bytes"]:
                        e["gbs"] = round(d["bytes"] / d["ms"] / 1e6, 1)
                        e["frac_of_hbm_peak"] = round(d["bytes"] / d["ms"] / 1e6 / HBM_PEAK_GBS, 4)
                        if d["n_large"]:
                            e["large_ops"] = dict(launches=d["n_large"], ms_per_step=round(d["ms_large"], 3), gbs=round(d["bytes_large"] / d["ms_large"] / 1e6, 1),
                                                  frac_of_hbm_peak=round(d["bytes_large"] / d["ms_large"] / 1e6 / HBM_PEAK_GBS, 4),
                                                  cache_warm_burst4=dict(ms_per_step=round(d["ms_large_burst"], 3), gbs=round(d["bytes_large"] / d["ms_large_burst"] / 1e6, 1),
                                                                         note="bursts of 4 identical launches / 4: launch overhead amortised, launches 2 - 4 re-read the "
                                                                              "memory-side cache - NOT an HBM fraction") if d["ms_large_burst"] else None,
                                                  note=">= 32 MB of algorithmic traffic per launch, each timed as ONE launch between two HIP events (includes ~5 us of event / "
                                                       "launch overhead); the rest of the family are latency-bound launches on the 16x16 ... 2x2 levels")
                    fams[k] = e
                dom = max((k for k in fam if fam[k]["flops"]), key=lambda k: fam[k]["ms"])
                dk = dominant_kernel_live(net, plan, L, heaviest)
                # the contract's roofline object describes the DOMINANT KERNEL (flops per launch / average launch duration against
                # the roof of the pipe it runs on, PMC traffic per launch); the step-level view moves to roofline.step
                step_view = roof
                roof = dict(bound="mfma", achieved=dk["achieved"], peak=dk["peak"], unit="TFLOP/s", frac=dk["frac"], traffic=dk["traffic"],
                            kernel=dk["kernel"], op=dk["op"], layer=dk["layer"], flops_per_launch=dk["flops_per_launch"],
                            avg_launch_ms=dk["avg_launch_ms"], algorithmic_bytes=dk["algorithmic_bytes"], peak_note=dk["peak_note"],
                            note="dominant kernel = the convolution launch that takes the most chip time of the step (duration x share of the CUs its grid "
                                 "occupies), re-timed live (20 launches between two HIP events on its launch stream); achieved = algorithmic fp32-equivalent FLOPs per launch / average launch duration; traffic = "
                                 "PMC-measured HBM bytes per launch of this kernel on this layer (profiles/), null when no committed PMC pass covers it")
                for k in ("fp16_mfma_tflops", "traffic_source", "traffic_table_git_blob", "sustained_ceiling_note"):
                    if k in dk:
                        roof[k] = dk[k]
                def grid_note(d_):
                    wg = int(L.uz_get_wgrad_target())
                    return ("this launch is cut into %d workgroups ON PURPOSE (uz_set_wgrad_target: its workgroups hold 472 - 508 of a SIMD's 512 registers "
                            "for the whole kernel, so nothing else starts on the CUs it occupies; with %d of 256 CUs the step is 3 %% faster, profiles/NOTES_r5.md) - "
                            "`frac` is against the WHOLE chip's peak; against the peak of the CUs it occupies: %.4f" % (wg, wg, d_["frac"] * 256.0 / wg))

                def is_reduced(d_):
                    return d_["op"] == "weight gradient" and d_["peak"] != PEAK_F32_MFMA_TFLOPS and L.uz_get_wgrad_target() < 256
                if is_reduced(dk):
                    roof["grid_note"] = grid_note(dk)
                if heaviest_full is not None and heaviest_full[1:3] != heaviest[1:3]:
                    d2 = dominant_kernel_live(net, plan, L, heaviest_full)
                    other = {k: d2[k] for k in ("kernel", "op", "layer", "avg_launch_ms", "achieved", "peak", "frac", "traffic", "algorithmic_bytes") if k in d2}
                    if is_reduced(d2):                  # the longest launch of the step, on a reduced grid: less chip time than the dominant one
                        other["grid_note"] = grid_note(d2)
                        roof["longest_launch_reduced_grid"] = other
                    elif is_reduced(dk):
                        roof["longest_full_grid_launch"] = other
                step_view.pop("traffic", None)
                roof["step"] = step_view
                roof["families"] = fams
                roof["dominant_family"] = dict(name=dom, **fams[dom])
            except Exception as e:                          # never lose the headline line to the per-family pass
                roof["families_error"] = str(e)[:200]
        b16 = getattr(getattr(net, "_cur", None), "b16_info", None) or {}
        store_b16 = bool(b16.get("buffers"))
        math_note = ("fp32 MFMA only (UZ_CONV_MATH=f32)" if conv_math() == "f32" else
                     "bf16 STORAGE + bf16 ARITHMETIC (UZ_CONV_MATH=bf16, UZ_STORE_B16=1): the volume's large tensors - %d activation buffers, %d gradient "
                     "buffers, the dy of %d units (%.2f GB less than fp32) - hold 2-byte bf16 elements (round to nearest even when written); the 3x3x3 "
                     "convolutions stage them as they are, one v_mfma_f32_32x32x16_bf16 product per MAC, fp32 accumulation; BatchNorm statistics fp32 / fp64 "
                     "of the stored values, parameters / gradients / optimiser state fp32; planes of 32 x 32 and below, the 1x1x1 heads' operands, the "
                     "in-plane interpolation stage, latent and loss tensors stay fp32" % (b16["buffers"], b16["grads"], b16["dy"], b16["bytes_saved"] / 1e9)
                     if (conv_math() == "bf16" and store_b16) else
                     "bf16 ARITHMETIC in the large 3x3(x3) convolutions (UZ_CONV_MATH=bf16): operands rounded to bf16 (round to nearest even) while they "
                     "are staged, one v_mfma_f32_32x32x16_bf16 product per MAC, fp32 accumulation; activations, gradients and BatchNorm statistics are "
                     "STORED in fp32 (bf16 storage is not built); small planes and 1x1 heads stay on the fp32 kernels" if conv_math() == "bf16" else
                     "fp32 in / fp32 out, fp32 accumulate everywhere; 3x3 layers the library routes to the split path (forward, data gradient AND "
                     "weight gradient; share in roofline.flop_share_by_pipe): operands scaled by a power of two and split into 2 fp16 pieces, 3 piece "
                     "products on the fp16 matrix pipe (22-bit operands: per layer within 2x the error of the fp32-MFMA kernels against fp64 (measured 0.5 - 0.8x on the heaviest layer); end to end at batch 32 the gradients' median error against fp64 is 1.6x the fp32 reference's own, logits within 1e-4, argmax bit-equal; tests/test_full_configs_gpu.py, tests/test_phiseg_gpu.py); other layers: fp32 MFMA")
        line = dict(metric=M["metric"], value=round(ips, 3 if vol else 2), unit=M.get("unit", "images/s"), n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(ms, 3), higher_is_better=True, scaling="strong" if args.strong else "weak",
                    vs_baseline=None, dtype=("bf16" if store_b16 else "bf16 arithmetic / f32 storage") if conv_math() == "bf16" else "f32", data="synthetic",
                    config=dict(workload=(M["workload"] % ("bf16" if store_b16 else "fp32")) if vol else M["workload"], batch_per_gpu=args.batch, global_batch=global_batch, parallelism=f"dp{world}",
                                graphs=not args.no_graphs, replay=(getattr(net, "replay_mode", "graph") if not args.no_graphs else "eager, one stream"),
                                lanes=getattr(plan, "n_lanes", None),
                                schedule=(dict(tuned_rounds=args.tune_schedule, **tuned, note="profile-guided lane schedule (Engine.tune_schedule) before the timed region")
                                          if tuned else dict(cost_model=os.environ.get("UZ_SCHED_COST", getattr(plan, "sched_cost", "alone")))),
                                final_loss=final_loss, conv_math=math_note, build=build_info(), env=uz_env(),
                                chain=getattr(plan, "chain_info", None) or None),
                    roofline=roof)
        if experiment:
            line["experiment"] = True
            line["config"]["experiment_reasons"] = experiment
        if world > 1:
            line["config"]["allreduce"] = "bucketed, overlapped with backward" if not args.no_overlap else "one blocking all-reduce after backward"
            sync = getattr(net, "_dp", None)
            if sync is not None:
                line["dp"] = dict(backend=sync.backend, nranks=sync.nranks(), control_plane="gloo (host): unique id, barriers, max of elapsed",
                                  communicators=1 if sync.backend == "rccl" else 0, launcher="self" if os.environ.get("UZ_BENCH_SELF_LAUNCHED") else "external",
                                  buckets_MB=[round(4 * (hi - lo) / 1e6, 1) for lo, hi in sync.buckets],
                                  exposed_allreduce_ms=sync.exposed_ms(), exposed_allreduce_ms_per_bucket=sync.exposed_ms_per_bucket())
                if line["dp"]["nranks"] != world:
                    raise SystemExit(f"data-parallel group reports {line['dp']['nranks']} ranks, the line says {world}")
        line.update(extra)
        if f32_leg is not None:
            line["fp32_mfma_only"] = f32_leg
        if not args.skip_cpu and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(args.model, args.cpu_batch)
            except Exception as e:                              # ... nor to the CPU leg
                line["cpu_baseline"] = dict(error=str(e)[:200])
        print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
