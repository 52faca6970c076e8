"""Data-parallel path on CPU: world_size-2 gloo process group, flat-gradient averaging, sharding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import unet_zoo_amd  # noqa: F401
from unet_zoo_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    # DP semantics: the averaged buffer equals the mean of the per-shard gradients (SURVEY 8e parity gate)
    dp.allreduce_mean_(flat)
    params = torch.full((8,), float(rank))
    dp.broadcast_(params, src=0)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), torch.cat([flat, params]).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_mean_world2_gloo(tmp_path):
    world, n = 2, 1000
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    ref = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    for a in got:
        assert np.allclose(a[:n], ref.numpy(), atol=1e-6)
        assert np.all(a[n:] == 0.0)                 # parameters broadcast from rank 0
    assert np.array_equal(got[0], got[1])           # replicas stay bitwise identical


def _resched_worker(rank, world, port, out_dir):
    """Engine.tune_schedule's rank-consistency step without a GPU: every rank 'measures' different per-op durations, the ranks agree on
    the element-wise maximum (dp.max_over_ranks), reschedule from it - and must end with the same tape order and lanes."""
    import json
    import random
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      UZ_LANES="3", UZ_SCHED_HEAVY="off", UZ_SCHED_COST="beside")
    dp.init_from_env(backend="gloo")
    from tests import _golden as G
    from unet_zoo_amd.models.phiseg import PHISeg
    _, meta = G.load("phiseg_small")
    net = PHISeg(1, 2, meta["filters"], image_size=(1, 64, 64), device="cpu")
    plan = net._build(2, 64, 64, True, True)
    rng = random.Random(1000 + rank)
    out = {"flags": dp.or_flags(1 << rank), "max": dp.max_over_ranks([float(rank), 10.0 - rank, 3.5])}
    for which, ops in (("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops)):
        prog = plan._program_order[which]
        mine = [rng.choice([3.0, 30.0, 300.0]) for _ in prog]
        agreed = dp.max_over_ranks(mine)
        for o, c in zip(prog, agreed):
            o["cost_us"] = c
        plan.reschedule(which)
        index = {id(o): k for k, o in enumerate(prog)}
        out[which] = [(index[id(o)], o["lane"]) for o in ops]
        out[which + "_mine"] = mine[:8]
    json.dump(out, open(os.path.join(out_dir, f"s{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_that_agree_on_the_measured_costs_end_with_one_schedule_world2_gloo(tmp_path):
    import json
    world = 2
    mp.spawn(_resched_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = (json.load(open(tmp_path / f"s{r}.json")) for r in range(world))
    assert a["flags"] == b["flags"] == 3 and a["max"] == b["max"] == [1.0, 10.0, 3.5]
    assert a["fwd_mine"] != b["fwd_mine"]                    # the ranks did measure different things
    for which in ("fwd", "bwd"):
        assert a[which] == b[which] and len(a[which]) > 50, which


def test_shard_bounds_cover_batch():
    for n, world in [(32, 8), (32, 3), (5, 8), (1, 2)]:
        spans = [dp.shard_bounds(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_single_process_is_a_noop():
    t = torch.arange(4.0)
    assert dp.allreduce_mean_(t.clone()).equal(t)


def test_gradient_buckets_follow_the_subnetworks():
    from unet_zoo_amd.models.phiseg import PHISeg
    from unet_zoo_amd.models.probabilistic_unet import ProbabilisticUnet
    from unet_zoo_amd.models.unet import Unet
    nf7 = [32, 64, 128, 192, 192, 192, 192]
    net = PHISeg(1, 2, nf7, image_size=(1, 128, 128), device="cpu")
    # round 6: byte-weighted slices of ~12 MB cut at tensor boundaries (rounds 2 - 5: one bucket per sub-network, 37.7 / 22.6 / 37.7 MB)
    b = dp.param_buckets(net._ptab)
    sizes = [hi - lo for lo, hi in b]
    target = 3 << 20
    assert len(b) == 7 and sum(sizes) == net._ptab.n_params == 24513330
    assert all(target <= n < target + 700000 for n in sizes[:-1]) and target // 2 <= sizes[-1] < 2 * target      # a slice ends at the first tensor boundary past the target
    assert b[0][0] == 0 and b[-1][1] == net._ptab.n_params and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    starts = {off for off in net._ptab.poff.values()} | {net._ptab.n_params}
    assert all(lo in starts and hi in starts for lo, hi in b)                      # never inside a tensor
    assert [hi - lo for lo, hi in dp.param_buckets(net._ptab, target_floats=1 << 30)] == [24513330]
    pu = ProbabilisticUnet(1, 2, nf7, latent_dim=6, no_convs_fcomb=3, device="cpu")
    bp = dp.param_buckets(pu._ptab)
    assert bp[0][0] == 0 and bp[-1][1] == pu._ptab.n_params and all(x[1] == y[0] for x, y in zip(bp, bp[1:])) and len(bp) == 5
    assert dp.param_buckets(Unet(1, 2, [32, 64, 128, 192], device="cpu")._ptab) == [(0, 2260194)]


def test_byte_weighted_buckets_on_the_headline_schedule():
    """VERDICT r5 item 3, on the headline plan's simulated backward schedule (the lane scheduler's own timeline, three lanes).  What the
    slices buy: the bucket that is final LAST holds one slice (<= 15 % of the bytes) instead of a sub-network (38.5 %), and a quarter of
    the bytes is final before 97 % of the tape.  What they do not buy, and why: 60 % of the bytes before 75 % of the tape -
    80 % of PHiSeg's parameters sit in the 16 x 16 ... 2 x 2 levels, whose weight gradients are the lanes' FILLERS; the three lanes are
    97 % busy, so running those ~100 launches earlier lengthens the tape by what it hides (measured in simulation, round 5: +11.6 %
    makespan for markers at 81 / 83 %).  The test pins the realised distribution so that a scheduler change that makes it worse fails."""
    from unet_zoo_amd.models.phiseg import PHISeg
    net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], image_size=(1, 128, 128), device="cpu")
    net.train()
    ref = net._build(32, 128, 128, True, True)
    end0 = max(g["sim_start"] + g["sim_cost"] for g in ref.dag[id(ref.bwd_ops)])
    net._plans.clear()
    net._dp = type("S", (), dict(overlap=True, buckets=dp.param_buckets(net._ptab)))()
    plan = net._build(32, 128, 128, True, True)
    ops, b = plan.bwd_ops, plan.grad_buckets
    dag = plan.dag[id(ops)]
    end = max(g["sim_start"] + g["sim_cost"] for g in dag)
    tot = sum(hi - lo for lo, hi in b)
    final = sorted((g["sim_start"] / end, (b[ops[g["first"]]["p"][0][1]][1] - b[ops[g["first"]]["p"][0][1]][0]) / tot)
                   for g in dag if ops[g["first"]]["code"] == "UZ_OP_EVENT_RECORD")
    assert len(final) == len(b) == 7
    assert final[-1][1] <= 0.15                                   # the exchange that trails the tape is one slice
    assert sum(f for t, f in final if t <= 0.97) >= 0.25
    assert final[0][0] <= 0.90
    assert end <= 1.01 * end0                                     # bucket markers and per-bucket reduction tables cost the tape < 1 %


def test_bucket_events_are_scheduled_behind_every_writer_of_their_bucket():
    """The data-parallel plan carries one UZ_OP_EVENT_RECORD per bucket; the lane scheduler must order it behind every
    op that writes a gradient of that bucket (brute-force check of tests/test_host_cpu.py) and may hoist it ahead of the
    rest of the tape - that hoisting is the overlap."""
    from tests.test_host_cpu import _check_lane_schedule
    from unet_zoo_amd.models.phiseg import PHISeg
    net = PHISeg(1, 2, [4, 8, 8, 8, 8, 8, 8], image_size=(1, 64, 64), device="cpu")
    net._dp = type("S", (), dict(overlap=True, buckets=dp.param_buckets(net._ptab, target_floats=3000)))()
    plan = net._build(2, 64, 64, True, True)
    ops = plan.bwd_ops
    marks = [k for k, o in enumerate(ops) if o["code"] == "UZ_OP_EVENT_RECORD"]
    assert len(marks) == len(plan.grad_buckets) >= 3
    _check_lane_schedule(plan, "bwd", ops)
    for k in marks:
        b = ops[k]["p"][0][1]
        lo, hi = plan.grad_buckets[b]
        writers = [j for j, o in enumerate(ops) for idx in plan._WRITES[o["code"]] if idx < len(o["p"])
                   for (space, a, c) in plan._resources(o["p"][idx]) if space == ("gflat",) and a < hi and lo < c]
        assert writers and max(writers) < k
    assert min(marks) < len(ops) - 3          # at least one bucket is final well before the tape ends
    # deferred reductions are cut per bucket: every table writes gradients of ONE bucket only (so that bucket's marker, and nobody
    # else's, waits for it), at most one table of a kind per bucket
    for code in ("UZ_OP_WGRAD_REDUCE_TABLE", "UZ_OP_CHAN_SUM_TABLE"):
        tabs = [o for o in ops if o["code"] == code]
        assert len(tabs) <= len(plan.grad_buckets)
        owners = []
        for o in tabs:
            offs = {plan.ptab.poff[k] for k in o["p"][1][1]}
            own = {b for b, (lo, hi) in enumerate(plan.grad_buckets) if any(lo <= x < hi for x in offs)}
            assert len(own) == 1, (code, own)
            owners.append(own.pop())
        assert len(set(owners)) == len(owners)
    assert any(o["code"] == "UZ_OP_WGRAD_REDUCE_TABLE" for o in ops)


def test_data_parallel_plan_without_tables_reduces_behind_every_layer(monkeypatch):
    from unet_zoo_amd.models.phiseg import PHISeg
    monkeypatch.setenv("UZ_DP_TABLES", "0")
    net = PHISeg(1, 2, [4, 8, 8, 8, 8, 8, 8], image_size=(1, 64, 64), device="cpu")
    net._dp = type("S", (), dict(overlap=True, buckets=dp.param_buckets(net._ptab, target_floats=3000)))()
    plan = net._build(2, 64, 64, True, True)
    assert not [o for o in plan.bwd_ops if o["code"] in ("UZ_OP_WGRAD_REDUCE_TABLE", "UZ_OP_CHAN_SUM_TABLE")]


def _sync_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dp.init_from_env(backend="gloo")
    from unet_zoo_amd.models.phiseg import PHISeg
    torch.manual_seed(0)
    net = PHISeg(1, 2, [4, 8, 8, 8, 8, 8, 8], image_size=(1, 64, 64), device="cpu")
    net.set_data_parallel(True)                              # gloo group -> torch backend of GradSync
    assert net._dp.backend == "torch" and net._dp.world == world
    net._ptab.pflat.fill_(float(rank))
    net._dp.broadcast_params()
    plan = net._build(2, 64, 64, True, True)
    g = torch.Generator().manual_seed(200 + rank)
    net._ptab.gflat.copy_(torch.randn(net._ptab.n_params, generator=g))
    net._dp.sync(plan)
    np.save(os.path.join(out_dir, f"s{rank}.npy"), torch.cat([net._ptab.gflat, net._ptab.pflat[:4]]).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_buckets_world2_gloo(tmp_path):
    """N>1 path of the product's gradient exchange (bucket partition, bucket order taken from the scheduled tape,
    averaging, parameter broadcast) with two gloo ranks: every rank ends with mean_r(g_r)."""
    world = 2
    port = _free_port()
    mp.spawn(_sync_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"s{r}.npy") for r in range(world)]
    n = got[0].size - 4
    ref = sum(torch.randn(n, generator=torch.Generator().manual_seed(200 + r)) for r in range(world)) / world
    for a in got:
        assert np.allclose(a[:n], ref.numpy(), atol=1e-6)
        assert np.all(a[n:] == 0.0)
    assert np.array_equal(got[0], got[1])


# ---------------------------------------------------------------------------------------------- bench.py --gpus N without a launcher
def _run_bench(extra_args, env_extra):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra_args, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                     # ONE JSON line, printed by rank 0 only
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start the ranks itself and report what the group saw (dry hook: the
    launcher, rendezvous, barrier / max-over-ranks timing and reporting code of the real run, a sleep instead of the step)."""
    d = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], dict(UZ_BENCH_DRY="1"))
    assert d["n_gpus"] == 2 and d["dp"]["nranks"] == 2 and d["dp"]["launcher"] == "self"
    assert d["config"]["global_batch"] == 64 and d["config"]["batch_per_gpu"] == 32 and d["scaling"] == "weak"
    s = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--strong"], dict(UZ_BENCH_DRY="1"))
    assert s["n_gpus"] == 2 and s["scaling"] == "strong" and s["config"]["global_batch"] == 32 and s["config"]["batch_per_gpu"] == 16


def test_bench_refuses_a_gpu_count_that_is_not_the_world_size():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UZ_BENCH_DRY="1", WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_bench_launcher_stops_the_survivors_when_a_rank_dies():
    """A rank that exits before the rendezvous would leave rank 0 waiting for it forever: the self-launcher must notice, stop the
    other ranks and fail (non-zero exit, message on stderr) instead of hanging until the driver's timeout."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(UZ_BENCH_DRY="1", UZ_BENCH_DRY_FAIL_RANK="1")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "rank 1 exited with code 3" in r.stderr, (r.returncode, r.stderr[-400:])
    assert time.time() - t0 < 200


def test_bench_launcher_deadline_stops_a_hung_rank():
    """A rank stuck inside communicator set-up never exits: the launcher's wall-clock deadline (UZ_BENCH_DEADLINE_S) must stop every
    rank and fail with each rank's exit code and stderr tail instead of waiting for the caller's timeout."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(UZ_BENCH_DRY="1", UZ_BENCH_DRY_HANG_RANK="1", UZ_BENCH_DEADLINE_S="20")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "no result after 20 s" in r.stderr and "rank 1 hangs on purpose" in r.stderr and "---- rank 0" in r.stderr, r.stderr[-600:]
    assert time.time() - t0 < 120
