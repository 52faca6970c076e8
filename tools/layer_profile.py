#!/usr/bin/env python3
"""Per-layer dispatch workload for rocprofv3 (kernel trace and the three PMC passes of tools/prof_layers.sh).

Builds the benchmark's PHiSeg plan, runs two training steps so that every buffer, bound slot and packed weight image holds real
values, picks the TOP heaviest convolution ops of the forward / backward tapes (ranked by a HIP-event timing of each op alone) and
then launches, per op:   one MARKER kernel (uz_axpy on one float - a kernel name that occurs nowhere else in this workload),
followed by REPS launches of exactly that tape op (uz_run_tape on a 1-op tape: the product's own call, its real views, bounds and
pre-packed weights) and a closing MARKER.  The dispatch stream is therefore   marker, op0 x REPS, marker, [prefix of op1], marker,
op1 x REPS, marker, ...   and tools/layer_table.py cuts the trace / counter CSVs at the marker PAIRS - keyed on dispatch ORDER, not
on grid size or kernel name.
Segment 0 is the FETCH_SIZE calibration (uz_absmax over a known byte count: one coalesced dword per lane).
usage: layer_profile.py <ops.json> [top] [reps]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bench import conv_dims, conv_flops, conv_bytes, conv_roof
from unet_zoo_amd import _ffi
from unet_zoo_amd.synthetic import synthetic_batch

out_path = sys.argv[1]
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 20
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
MODEL = os.environ.get("UZ_PROFILE_MODEL", "phiseg")
B = 32
net = bench.build(MODEL); net.train()
x, m, _ = synthetic_batch(B)
x, m = torch.from_numpy(x).cuda(), torch.from_numpy(m).cuda()
for _ in range(2):
    net.forward(x) if MODEL == "unet" else net.forward(x, m)
    net.loss(m).backward()
torch.cuda.synchronize()
plan = net._cur
L = _ffi.lib()
st = C.c_void_p(net._stream())


def one_op(which, k):
    arr, _n = plan.tapes[which]
    return (type(arr[0]) * 1)(arr[k])


def run_group_prefix(which, k):
    """The ops of the same scheduling group in front of op k, once: a backward convolution reads its layer's dy from the lane's
    SCRATCH, which after a full step holds some other layer's dy (values beyond this op's magnitude bound: clamped, flagged, and
    not what the op sees in a step).  Re-running the BatchNorm backward of the group puts the right tensor there."""
    ops = plan.fwd_ops if which == "fwd" else plan.bwd_ops
    arr, _n = plan.tapes[which]
    j = k
    while j > 0 and ops[j - 1]["gid"] == ops[k]["gid"]:
        j -= 1
    if j < k:
        sub = (type(arr[0]) * (k - j))(*[arr[i] for i in range(j, k)])
        _ffi.check(L.uz_run_tape(sub, k - j, st), "group prefix")


cands = []
fixed = os.environ.get("UZ_PROFILE_OPS")            # the PMC passes profile exactly the op list the trace pass ranked
if fixed and os.path.exists(fixed):
    top = [(o["isolated_ms_hip_events"], o["tape"], o["index"]) for o in json.load(open(fixed))["ops"]]
for which, ops in ((("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops)) if not (fixed and os.path.exists(fixed)) else ()):
    for k, o in enumerate(ops):
        if conv_dims(o) is None:
            continue
        tape = one_op(which, k)
        run_group_prefix(which, k)
        _ffi.check(L.uz_run_tape(tape, 1, st), "op")        # untimed first launch (kernel attribute calls, code object residency)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _ffi.check(L.uz_run_tape(tape, 1, st), "op"); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        cands.append((best, which, k))
if cands:
    cands.sort(reverse=True)
    top = cands[:TOP]

marker_buf = torch.zeros(4, device="cuda")
cal = torch.randn(256 * 1024 * 1024, device="cuda")                 # 1 GiB calibration read: four times the Infinity Cache, so no repetition finds it resident
flush = torch.empty(128 * 1024 * 1024, device="cuda") if os.environ.get("UZ_PROFILE_FLUSH") == "1" else None      # 512 MiB
slot = torch.zeros(256, device="cuda")
meta = dict(model=MODEL, batch=B, reps=REPS, marker="axpy_k", calibration=dict(kernel="absmax_view_kernel", known_read_bytes=cal.numel() * 4), ops=[])
torch.cuda.synchronize()


def marker():
    _ffi.check(L.uz_axpy(marker_buf.data_ptr(), marker_buf.data_ptr(), C.c_float(0.0), 1, st), "marker")


def cold():
    """PMC passes only (UZ_PROFILE_FLUSH=1): overwrite 512 MiB between two launches, so that every launch starts with none of its
    operands in the 256 MiB Infinity Cache - the traffic then is what the kernel pulls from HBM when nothing helps it (an upper
    bound of the in-step figure; FETCH_SIZE shrinks by a third when the same operands are re-read from the cache, measured on the
    calibration kernel).  The flush kernel (zero_k) is excluded from the per-op sums by name."""
    if flush is not None:
        _ffi.check(L.uz_zero_f32(flush.data_ptr(), flush.numel(), st), "flush")


marker()
for _ in range(REPS):
    cold()
    _ffi.check(L.uz_absmax(cal.data_ptr(), cal.numel(), slot.data_ptr(), st), "calibration")
marker()
for ms, which, k in top:
    o = (plan.fwd_ops if which == "fwd" else plan.bwd_ops)[k]
    kind, cin, cout, n, h, w, ks = conv_dims(o)
    roof = conv_roof(o, L)
    meta["ops"].append(dict(tape=which, index=k, op={0: "forward", 1: "data gradient", 2: "weight gradient"}[kind], kind=kind,
                            layer=f"{ks}x{ks} {cin}->{cout} @ {n}x{h}x{w}", cin=cin, cout=cout, n=n, h=h, w=w, ks=ks,
                            flops=conv_flops(o), algorithmic_bytes=conv_bytes(o), roof_tflops=roof, isolated_ms_hip_events=round(ms, 4)))
    tape = one_op(which, k)
    run_group_prefix(which, k)
    marker()
    for _ in range(REPS):
        cold()
        _ffi.check(L.uz_run_tape(tape, 1, st), "op")
    marker()                                             # a segment is [marker, launches, marker]: the next op's group prefix stays outside
torch.cuda.synchronize()
json.dump(meta, open(out_path, "w"), indent=1)
print("profiled", len(top), "ops; heaviest", meta["ops"][0]["layer"], meta["ops"][0]["op"], meta["ops"][0]["isolated_ms_hip_events"], "ms")
