#!/usr/bin/env python3
"""Headline benchmark: images/sec of one PHiSeg-7/5 training step (forward + loss + backward +
gradient all-reduce + Adam) at 128x128, batch 32 per GPU, fp32, on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement).  Workload = BASELINE.json
configs[3] (PHiSeg 7 resolution / 5 latent levels, filters 32,64,128,192,192,192,192), synthetic
LIDC-like inputs, random-init weights, inputs resident in HBM before the timed region.

Extra objects on the line:
  roofline      step-level fp32-MFMA roofline exactly as BASELINE.md section 2 defines it
                (achieved TFLOP/s = images/s x 100.36 GFLOP/image; peak 157.3 TFLOP/s), plus
                `dominant_kernel`: the single heaviest kernel (3x3 conv 224->128 at 32x128x128) timed
                live with HIP events on its launch stream - algorithmic FLOPs per launch / average
                launch duration - with its PMC-measured HBM bytes per launch as `traffic`;
                `families`: summed algorithmic FLOPs (or bytes) / summed duration of every kernel
                family of the step, each op bracketed by HIP events; `dominant_family` = the heaviest.
  cpu_baseline  the CPU oracle (a functional torch restatement of the reference graph = "port")
                timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FILTERS = [32, 64, 128, 192, 192, 192, 192]
GFLOP_PER_IMAGE = 100.36          # BASELINE.md section 2 (fwd+bwd, measured from the reference graph)
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md chip-level parameters
GB_PER_IMAGE_UNFUSED = 1.077
GB_PER_STEP_FIXED = 0.294
HBM_PEAK_GBS = 8000.0


def conv_flops(op, codes):
    """Algorithmic FLOPs of one conv-family tape op (2 * N*H*W * Cin*Cout*k*k)."""
    i = op["i"]
    c = op["code"]
    if c == "UZ_OP_CONV_FWD":
        cin, cout, n, h, w, ks = i[0], i[2], i[4], i[5], i[6], i[7]
    elif c == "UZ_OP_CONV_BWD_DATA":
        cout, cin, n, h, w, ks = i[0], i[2], i[4], i[5], i[6], i[7]
    elif c == "UZ_OP_CONV_BWD_WEIGHT":
        cin, cout, n, h, w, ks = i[0], i[2], i[4], i[5], i[6], i[7]
    else:
        return 0.0
    return 2.0 * n * h * w * cin * cout * ks * ks


def op_bytes(op, plan):
    """Algorithmic HBM bytes of a streaming (non-conv) op: every tensor argument read or written once."""
    c, i = op["code"], op["i"]
    f4 = 4.0
    if c == "UZ_OP_BN_RELU_FWD":
        C, N, H, W, training = i[0], i[3], i[4], i[5], i[6]
        big = N * H * W > 8192
        return f4 * C * N * H * W * ((3 if big else 2) if training else 2)
    if c == "UZ_OP_BN_RELU_BWD":
        C, N, H, W = i[1], i[4], i[5], i[6]
        big = N * H * W > 8192
        return f4 * C * N * H * W * (5 if big else 3)
    if c in ("UZ_OP_AVGPOOL_FWD", "UZ_OP_AVGPOOL_BWD"):
        C, N, H, W = i[0], i[3], i[4], i[5]
        return f4 * C * N * H * W * 1.25
    if c in ("UZ_OP_BILINEAR_FWD", "UZ_OP_BILINEAR_BWD"):
        C, N, H, W = i[0], i[3], i[4], i[5]
        return f4 * C * N * H * W * 5
    if c in ("UZ_OP_NEAREST_FWD", "UZ_OP_NEAREST_BWD"):
        C, N, H, W, f = i[0], i[3], i[4], i[5], i[6]
        return f4 * C * N * H * W * (1 + f * f)
    return 0.0


FAMILY = {"UZ_OP_CONV_FWD": "conv_fwd_mfma", "UZ_OP_CONV_BWD_DATA": "conv_dgrad_mfma", "UZ_OP_CONV_BWD_WEIGHT": "conv_wgrad_mfma",
          "UZ_OP_BN_RELU_FWD": "bn_relu_fwd", "UZ_OP_BN_RELU_BWD": "bn_relu_bwd",
          "UZ_OP_AVGPOOL_FWD": "resample", "UZ_OP_AVGPOOL_BWD": "resample", "UZ_OP_BILINEAR_FWD": "resample",
          "UZ_OP_BILINEAR_BWD": "resample", "UZ_OP_NEAREST_FWD": "resample", "UZ_OP_NEAREST_BWD": "resample"}


def profile_families(net, plan, reps=3):
    """Live per-family timing: replay the fwd / bwd tapes one op at a time, each bracketed by HIP
    events on the launch stream (torch.cuda.Event records on torch's current stream, which IS the
    stream the tape is launched on)."""
    import ctypes as C
    from unet_zoo_amd import _ffi
    L = _ffi.lib()
    stream = C.c_void_p(net._stream())
    fam = {}
    for which, ops in (("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops)):
        arr, n = plan.tapes[which]
        for k in range(n):
            name = FAMILY.get(ops[k]["code"], "other")
            one = (type(arr[0]) * 1)(arr[k])
            best = None
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _ffi.check(L.uz_run_tape(one, 1, stream), "profile op")
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1)
                best = ms if best is None else min(best, ms)
            d = fam.setdefault(name, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            d["ms"] += best
            d["flops"] += conv_flops(ops[k], None)
            d["bytes"] += op_bytes(ops[k], plan)
            d["launches"] += 1
    return fam


PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak
SPLIT_PRODUCTS = 6                # bf16 piece products per fp32 product in conv_split.hip


def conv_math():
    return os.environ.get("UZ_CONV_MATH", "default")


def dominant_kernel_live(dev, reps=20):
    """The single heaviest kernel of the step - the 3x3 forward convolution 224 -> 128 at 32 x 128 x 128 -
    timed live with HIP events on the stream it is launched on.  Algorithmic FLOPs per launch =
    2*N*H*W*Cin*Cout*9.  In the default mode this layer runs on conv_split_kernel<2> (three bf16 pieces
    per fp32 operand, six piece products on the bf16 matrix pipe, fp32 accumulate): its roof is the
    dense bf16 MFMA peak divided by the six products.  With UZ_CONV_MATH=f32 it runs on
    conv_mfma_kernel<3,2,2,false> against the fp32 MFMA peak.  HBM bytes per launch come from the
    committed PMC passes (profiles/r1_pmc_traffic.json: FETCH_SIZE / WRITE_SIZE, calibrated)."""
    import ctypes as C
    from unet_zoo_amd import _ffi
    L = _ffi.lib()
    Cin, Cout, N, H, W = 224, 128, 32, 128, 128
    x = torch.randn(N, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    y = torch.empty(N, Cout, H, W, device=dev)
    wsb = L.uz_conv_workspace(Cin, Cout, N, H, W, 3)
    ws = torch.zeros(wsb // 4 + 64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch():
        _ffi.check(L.uz_conv_fwd(x.data_ptr(), Cin, Cin, w.data_ptr(), None, y.data_ptr(), Cout, Cout, N, H, W, 3, 0,
                                 ws.data_ptr(), wsb, st), "conv_fwd")
    launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * N * H * W * Cin * Cout * 9
    split = conv_math() != "f32"
    peak = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS if split else PEAK_F32_MFMA_TFLOPS
    out = dict(kernel="conv_split_kernel<2> (+ pack_weights_kernel)" if split else "conv_mfma_kernel<3,2,2,false>",
               layer="3x3 224->128 @ 32x128x128", flops_per_launch=flops,
               avg_launch_ms=round(ms, 4), achieved=round(flops / ms / 1e9, 2), peak=round(peak, 1), unit="TFLOP/s",
               frac=round(flops / ms / 1e9 / peak, 4), traffic=None,
               peak_note=("dense bf16 MFMA peak 2500 TFLOP/s / 6 piece products per fp32 product" if split else "fp32 MFMA peak"))
    if split:
        out["bf16_mfma_tflops"] = round(SPLIT_PRODUCTS * flops / ms / 1e9, 1)
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
        k = pmc["kernels"]["conv_split_kernel<2> (forward)" if split else "conv_mfma_kernel<3,2,2,false> (forward)"]
        out.update(traffic=k["hbm_bytes"], algorithmic_bytes=k["algorithmic_bytes"], traffic_source="profiles/r1_pmc_traffic.json")
    except Exception:
        pass
    return out


def fp32_only_leg(args):
    """Same benchmark in a child process with UZ_CONV_MATH=f32 (every convolution on the fp32 MFMA kernels)."""
    import subprocess
    env = dict(os.environ, UZ_CONV_MATH="f32")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--batch", str(args.batch), "--skip-cpu", "--no-profile", "--no-f32-leg"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return dict(value=d["value"], unit=d["unit"], ms_per_step=d["ms_per_step"], frac=d["roofline"]["frac"],
                    note="identical run with UZ_CONV_MATH=f32: all convolutions on v_mfma_f32_32x32x2_f32")
    except Exception as e:                                   # never fail the headline line because of the extra leg
        return dict(error=str(e)[:200])


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


def cpu_baseline(batch, budget_s=25.0):
    """CPU oracle timed on the host cores on a bounded sample (about `budget_s` seconds of CPU work).
    Only this leg of bench.py imports oracle/."""
    import oracle
    from unet_zoo_amd.models.phiseg import phiseg_spec
    cores = usable_cores()
    threads = min(cores, 32)                  # oneDNN / OpenMP stop scaling (and thrash) far below 256 threads
    torch.set_num_threads(threads)
    sd = oracle.deterministic_state_dict(phiseg_spec(1, 2, FILTERS), seed=3)
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
        x, mask, eps = oracle.synthetic_batch(batch, 128, 128, seed=100 + step, eps_shapes=shapes + shapes)
        e = [torch.from_numpy(a) for a in eps]
        t0 = time.perf_counter()
        out = oracle.phiseg_forward(leaves, torch.from_numpy(x), torch.from_numpy(mask), dict(posterior=e[:5], prior=e[5:]))
        total, _ = oracle.phiseg_loss(out, torch.from_numpy(mask))
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
                sample=f"CPU oracle (functional torch fp32 restatement of the reference graph), PHiSeg 7/5 128x128 batch {batch}, "
                       f"{len(timed)} timed step(s){' after 1 warm-up' if len(times) > 1 else ' (warm-up only: budget exhausted)'}, "
                       f"fwd+loss+bwd+Adam, {threads} threads of {cores} usable cores, {sec:.2f} s/step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (weak scaling)")
    ap.add_argument("--no-graphs", action="store_true", help="launch kernels eagerly instead of hipGraph replay")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-family HIP-event pass")
    ap.add_argument("--skip-cpu", action="store_true", help="skip the CPU-oracle baseline leg")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the extra fp32-MFMA-only measurement (child process)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Test hooks (not used by the driver): UZ_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and UZ_BENCH_BACKEND=gloo
    # replaces RCCL, so that the multi-rank code path can be exercised on a one-GPU box.
    if os.environ.get("UZ_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("UZ_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from unet_zoo_amd.models.phiseg import PHISeg
    from unet_zoo_amd.synthetic import synthetic_batch
    from unet_zoo_amd.optim import FusedAdam

    torch.manual_seed(1234)          # same initial weights on every rank (DP replicas)
    net = PHISeg(input_channels=1, num_classes=2, num_filters=FILTERS, latent_levels=5, image_size=(1, 128, 128))
    net.train()
    if world > 1:
        dist.broadcast(net._ptab.pflat, src=0)
        net.set_data_parallel(True)
    if not args.no_graphs:
        net.enable_graphs(True)
    opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)        # train_model.py:49
    x, mask, _ = synthetic_batch(args.batch, 128, 128, seed=20201004 + rank)
    dev = torch.device("cuda", local_rank)
    x, mask = torch.from_numpy(x).to(dev), torch.from_numpy(mask).to(dev)

    def step():
        net.forward(x, mask, training=True)
        loss = net.loss(mask)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(max(args.warmup, 3 if not args.no_graphs else 1)):
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
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss.detach())

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        ips = args.batch * world * args.steps / elapsed
        per_gpu = ips / world
        roof = dict(bound="mfma", achieved=round(per_gpu * GFLOP_PER_IMAGE / 1e3, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                    frac=round(per_gpu * GFLOP_PER_IMAGE / 1e3 / PEAK_F32_MFMA_TFLOPS, 4), traffic=None,
                    hbm_fraction=round((per_gpu * GB_PER_IMAGE_UNFUSED + (per_gpu / args.batch) * GB_PER_STEP_FIXED) / HBM_PEAK_GBS, 4),
                    note="step-level, per GPU: images/s x 100.36 GFLOP/image over the 157.3 TFLOP/s fp32-MFMA peak (BASELINE.md section 2); "
                         "hbm_fraction uses the unfused-graph bytes and cannot exceed ~0.21 in fp32")
        if not args.no_profile and world == 1:
            fam = profile_families(net, net._cur)
            fams = {}
            for k, d in fam.items():
                e = dict(ms_per_step=round(d["ms"], 3), launches=d["launches"])
                if d["flops"]:
                    e["tflops"] = round(d["flops"] / d["ms"] / 1e9, 2)
                    e["frac_of_mfma_peak"] = round(d["flops"] / d["ms"] / 1e9 / PEAK_F32_MFMA_TFLOPS, 4)
                elif d["bytes"]:
                    e["gbs"] = round(d["bytes"] / d["ms"] / 1e6, 1)
                    e["frac_of_hbm_peak"] = round(d["bytes"] / d["ms"] / 1e6 / HBM_PEAK_GBS, 4)
                fams[k] = e
            dom = max((k for k in fam if fam[k]["flops"]), key=lambda k: fam[k]["ms"])
            roof["families"] = fams
            roof["dominant_family"] = dict(name=dom, **fams[dom])
            roof["dominant_kernel"] = dominant_kernel_live(dev)
            roof["traffic"] = roof["dominant_kernel"]["traffic"]
        line = dict(metric="images/sec fwd+bwd PHiSeg-7 128x128 bs32", value=round(ips, 2), unit="images/s", n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(ms, 3), higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype="f32", data="synthetic",
                    config=dict(workload="PHiSeg 7 resolution / 5 latent levels, filters 32-64-128-192x4, 1x128x128, fwd+loss+bwd+Adam",
                                batch_per_gpu=args.batch, global_batch=args.batch * world, parallelism=f"dp{world}",
                                graphs=not args.no_graphs, final_loss=final_loss,
                                conv_math=("fp32 MFMA only (UZ_CONV_MATH=f32)" if conv_math() == "f32" else
                                           "fp32 in / fp32 out; large 3x3 layers (fwd, dgrad): operands split exactly into 3 bf16 pieces, "
                                           "6 piece products on the bf16 matrix pipe, fp32 accumulate (error vs fp64 1.7x the fp32-MFMA "
                                           "kernel's, logits 2.6e-5 from the reference); other layers and all weight gradients: fp32 MFMA")),
                    roofline=roof)
        if world == 1 and not args.no_f32_leg and conv_math() != "f32":
            line["fp32_mfma_only"] = fp32_only_leg(args)
        if not args.skip_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.cpu_batch)
        print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
