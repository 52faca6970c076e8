"""Does the host keep ahead of the device in the replayed PHiSeg training step, and where are the seams between its calls?

Three measurements on the bench's own step (forward graph, loss, zero_grad, backward graph, Adam), un-profiled:
  A  free running: host wall time spent inside each call, per step (no synchronisation inside the loop);
  B  device-side seams: events between the calls give the device time of each section of a free-running step;
  C  each call alone: synchronise, call, synchronise - pure enqueue cost on an idle queue and the call's isolated device time.
If sum(A) per step is well below the step time the host is ahead and the seams are the device's; if sum(C device) is below the
free-running step, time is lost BETWEEN the calls.  Usage: python tools/host_runahead.py [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    from unet_zoo_amd.optim import FusedAdam
    from unet_zoo_amd.synthetic import synthetic_batch
    torch.manual_seed(1234)
    net = bench.build("phiseg")
    net.train()
    net.enable_graphs(True)
    opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)
    dev = torch.device("cuda", 0)
    x, mask, _ = synthetic_batch(32, 128, 128, seed=20201004)
    x, mask = torch.from_numpy(x).to(dev), torch.from_numpy(mask).to(dev)
    names = ["forward", "loss", "zero_grad", "backward", "adam"]

    def calls():
        st = {}
        return [lambda: net.forward(x, mask, training=True),
                lambda: st.__setitem__("loss", net.loss(mask)),
                lambda: opt.zero_grad(),
                lambda: st["loss"].backward(),
                lambda: opt.step()]

    cs = calls()
    for _ in range(5):
        for c in cs:
            c()
    torch.cuda.synchronize()

    # A: free running
    host = [0.0] * 5
    t0 = time.perf_counter()
    for _ in range(steps):
        for i, c in enumerate(cs):
            a = time.perf_counter()
            c()
            host[i] += time.perf_counter() - a
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    out = {"A_free_running": {"ms_per_step": round(1e3 * t_all / steps, 3), "host_enqueue_ms_per_step": round(1e3 * t_enq / steps, 3),
                              "host_ms_by_call": {n: round(1e3 * h / steps, 3) for n, h in zip(names, host)}}}

    # B: device-side sections of a free-running step
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        ev[s][0].record()
        for i, c in enumerate(cs):
            c()
            ev[s][i + 1].record()
    torch.cuda.synchronize()
    t_b = time.perf_counter() - t0
    sec = [0.0] * 5
    seam = 0.0
    for s in range(steps):
        for i in range(5):
            sec[i] += ev[s][i].elapsed_time(ev[s][i + 1])
        if s:
            seam += ev[s - 1][5].elapsed_time(ev[s][0])
    out["B_device_sections"] = {"ms_per_step": round(1e3 * t_b / steps, 3), "device_ms_by_call": {n: round(v / steps, 3) for n, v in zip(names, sec)},
                                "between_steps_ms": round(seam / max(steps - 1, 1), 4)}

    # C: every call alone
    hostc, devc = [0.0] * 5, [0.0] * 5
    for _ in range(steps):
        for i, c in enumerate(cs):
            torch.cuda.synchronize()
            a = time.perf_counter()
            c()
            b = time.perf_counter()
            torch.cuda.synchronize()
            d = time.perf_counter()
            hostc[i] += b - a
            devc[i] += d - a
    out["C_isolated"] = {"host_ms_by_call": {n: round(1e3 * h / steps, 3) for n, h in zip(names, hostc)},
                         "call_to_idle_ms": {n: round(1e3 * h / steps, 3) for n, h in zip(names, devc)},
                         "sum_call_to_idle_ms": round(1e3 * sum(devc) / steps, 3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
