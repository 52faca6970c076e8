"""Worker of tests/test_dp_gpu.py: launched with torch.distributed.run, WORLD_SIZE ranks all on cuda:0 (gloo),
checks the data-parallel gradient path of the native PHISeg on the device:
  1. after the first loss.backward() every rank holds the SAME flat gradient buffer, equal to the mean of the
     ranks' local gradients (same weights, same injected noise, different data per rank);
  2. after two more steps (hipGraph capture + replay with the all-reduce between backward and Adam) the
     parameters are still bit-identical on every rank."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle                                              # noqa: E402  (test infrastructure: noise shapes only)
from unet_zoo_amd.models.phiseg import PHISeg            # noqa: E402
from unet_zoo_amd.optim import FusedAdam                 # noqa: E402
from unet_zoo_amd.synthetic import synthetic_batch       # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    filters = [8, 16, 16, 16, 16, 16, 16]
    B, HW = 4, 64
    x, mask, _ = synthetic_batch(B, HW, HW, seed=100 + rank)
    x, mask = torch.from_numpy(x).to(dev), torch.from_numpy(mask).to(dev)
    shapes = oracle.phiseg_eps_shapes(B, HW, HW)
    eps = [torch.full(tuple(s), 0.1 * (k + 1), device=dev) for k, s in enumerate(list(shapes) + list(shapes))]

    def make(dp):
        torch.manual_seed(7)
        net = PHISeg(1, 2, filters, latent_levels=5, image_size=(1, HW, HW))
        net.train()
        dist.broadcast(net._ptab.pflat, src=0)
        if dp:
            net.set_data_parallel(True)
        net.enable_graphs(True)
        return net, FusedAdam(net, lr=1e-3, weight_decay=1e-5)

    def step(net, opt):
        net.forward(x, mask, training=True, eps=eps)
        loss = net.loss(mask)
        net.zero_grad()
        loss.backward()
        g = net._ptab.gflat.clone()
        opt.step()
        return g

    net0, opt0 = make(False)
    g_local = step(net0, opt0)
    mean = g_local.clone()
    dist.all_reduce(mean)
    mean /= world
    net1, opt1 = make(True)
    g_dp = step(net1, opt1)
    dev_rel = float((g_dp - mean).abs().max() / (mean.abs().max() + 1e-30))
    for _ in range(2):
        step(net1, opt1)
    ref_g, ref_p = g_dp.clone(), net1._ptab.pflat.clone()
    dist.broadcast(ref_g, src=0)
    dist.broadcast(ref_p, src=0)
    same = torch.equal(ref_g, g_dp) and torch.equal(ref_p, net1._ptab.pflat)
    print(f"rank {rank}: identical_across_ranks={same} rel_dev_from_mean_of_local={dev_rel:.3e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if (same and dev_rel < 1e-5) else 1)


if __name__ == "__main__":
    main()
