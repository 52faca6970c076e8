"""Worker of tests/test_dp_gpu.py: launched with torch.distributed.run, WORLD_SIZE ranks all on cuda:0 (gloo - RCCL refuses
two ranks on one device), checks the data-parallel gradient path of the native PHISeg on the device against the ORACLE:
  1. after the first loss.backward() every rank holds the SAME flat gradient buffer, and it equals
     mean_r(oracle_gradients(shard_r)) computed by the CPU oracle per shard (SURVEY.md 8e parity gate);
  2. after two more steps (hipGraph capture + replay, bucketed exchange between backward and Adam), a profile-guided re-scheduling of
     the tapes (Engine.tune_schedule: one schedule and one bucket order on all ranks) and two steps under it, the gradients and
     parameters are still bit-identical on every rank."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle                                              # noqa: E402  (test infrastructure: the checker)
from tests import _golden as G                           # noqa: E402
from unet_zoo_amd.models.phiseg import PHISeg, phiseg_spec  # noqa: E402
from unet_zoo_amd.optim import FusedAdam                 # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    filters = [8, 16, 16, 16, 16, 16, 16]
    B, HW = 4, 128
    sd0 = oracle.deterministic_state_dict(phiseg_spec(1, 2, filters), seed=21)
    shapes = oracle.phiseg_eps_shapes(B, HW, HW)

    seed0 = int(os.environ.get("UZ_DP_TEST_SEED", "900"))

    def shard(r):
        return oracle.synthetic_batch(B, HW, HW, seed=seed0 + r, eps_shapes=shapes + shapes)

    x, mask, eps = shard(rank)
    xd, md = torch.from_numpy(x).to(dev), torch.from_numpy(mask).to(dev)
    ed = [torch.from_numpy(e).to(dev) for e in eps]
    net = PHISeg(1, 2, filters, latent_levels=5, image_size=(1, HW, HW))
    net.load_state_dict(sd0)
    net.train()
    net.set_data_parallel(True)                            # gloo group -> GradSync's torch backend, same buckets / order
    net._dp.broadcast_params()
    net.enable_graphs(True)
    opt = FusedAdam(net, lr=1e-3, weight_decay=1e-5)

    def step():
        net.forward(xd, md, training=True, eps=ed)
        loss = net.loss(md)
        net.zero_grad()
        loss.backward()
        g = net._ptab.gflat.clone()
        opt.step()
        return g

    g_dp = step()
    # oracle: mean over the ranks' shards of the per-shard gradients (every rank computes all shards on the CPU)
    ref = None
    for r in range(world):
        xr, mr, er = shard(r)
        lv = G.leaves(sd0)
        e = [torch.from_numpy(a) for a in er]
        out = oracle.phiseg_forward(lv, torch.from_numpy(xr), torch.from_numpy(mr), dict(posterior=e[:5], prior=e[5:]))
        total, _ = oracle.phiseg_loss(out, torch.from_numpy(mr))
        total.backward()
        gr = {k: v.grad for k, v in lv.items() if v.requires_grad}
        ref = gr if ref is None else {k: (None if v is None else v + gr[k]) for k, v in ref.items()}
    noise = G.bn_shadowed_biases(ref.keys())
    worst, devs = 0.0, []
    for k, v in ref.items():
        if v is None or k in noise:
            continue
        v = v / world
        mine = net._ptab.gview(k).cpu() if False else g_dp[net._ptab.poff[k]:net._ptab.poff[k] + v.numel()].view(v.shape).cpu()
        devs.append(float((mine - v).abs().max() / (1e-3 + v.abs().max())))
        worst = max(worst, devs[-1])
    for _ in range(2):
        step()
    if net.replay_mode == "lanes":
        # profile-guided schedule under data parallelism: the ranks agree on every measured number (dp.max_over_ranks), so they end with
        # ONE schedule and one bucket exchange order - ranks that tuned on their own would exchange different buckets with each other
        res = net.tune_schedule(step, rounds=2, samples=2, validate=2)
        plan = net._cur
        mine = [net._dp.order(plan), [o["lane"] for o in plan.bwd_ops]]
        every = [None] * world
        dist.all_gather_object(every, mine)
        assert all(e == every[0] for e in every), "ranks hold different schedules after tune_schedule()"
        print(f"rank {rank}: tuned {res['kept']} {res['step_ms']} bucket order {mine[0]}", flush=True)
        g_dp = step()
        step()
    ref_g, ref_p = g_dp.clone(), net._ptab.pflat.clone()
    dist.broadcast(ref_g, src=0)
    dist.broadcast(ref_p, src=0)
    same = torch.equal(ref_g, g_dp) and torch.equal(ref_p, net._ptab.pflat)
    print(f"rank {rank}: identical_across_ranks={same} worst_rel_dev_from_mean_of_oracle_shard_grads={worst:.3e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    # Gates.  What this test is about - the exchange - fails by O(1): a shard left out or a sum instead of the mean moves EVERY tensor by
    # 50 - 100 %, a single mis-bucketed layer or an overwritten regulariser share moves ITS tensors by 3 - 10 %.  Rounding does not: the
    # median tensor sits at 1e-4 of its largest entry (gate 1e-3) and the worst at 4e-3 - 6e-3 (gate 2e-2) - on THIS
    # data seed.  The seed matters: a pixel whose pre-activation is zero to an ulp can carry the largest upstream gradient of its
    # tensor, and whether its ReLU mask is 0 or 1 then moves a weight gradient by 3 % (seeds 100 and 500 have such a pixel: worst
    # 2.7e-2 with 1 - 2 tensors beyond 2e-2, tools/dp_seed_scan.sh; seeds 300 / 700 / 900: 1.6e-2 / 4.4e-3 / 5.6e-3, none) - so the
    # test uses a seed without one instead of a gate wide enough to hide a real fault.
    med = sorted(devs)[len(devs) // 2]
    p90 = sorted(devs)[(9 * len(devs)) // 10]
    above = sum(d >= 2e-2 for d in devs)
    print(f"rank {rank}: median_rel_dev={med:.3e} p90={p90:.3e} tensors_above_2e-2={above} seed={seed0}", flush=True)
    if os.environ.get("UZ_DP_TEST_GATE") == "knife_edge":
        # a gate that does not pass by seed selection.  On a seed WITH a knife-edge ReLU pixel (100, 500) the distribution is
        # gated instead of the single worst tensor: a mis-bucketed layer moves its weight, bias and BatchNorm tensors - and through the
        # optimiser step everything downstream - so it shows as MANY tensors beyond 2e-2 and a moved 90th percentile; a knife-edge pixel
        # moves the one or two tensors it feeds (measured 2.7e-2) and nothing else.
        sys.exit(0 if (same and med < 1e-3 and p90 < 5e-3 and above <= 2 and worst < 6e-2) else 1)
    sys.exit(0 if (same and worst < 2e-2 and med < 1e-3) else 1)


if __name__ == "__main__":
    main()
