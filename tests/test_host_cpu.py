"""CPU-tier checks of the host side: the C-ABI library loads and exports every symbol the header
declares, the state_dict surface equals the reference's (golden key lists), plans build, and the
product path refuses to run without a GPU (no CPU fallback)."""
import json
import os

import pytest
import torch

import unet_zoo_amd  # noqa: F401
from unet_zoo_amd import _ffi
from tests import _golden as G


def test_library_exports_every_declared_symbol():
    L = _ffi.lib()
    names = _ffi.header_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), n
    assert L.uz_version() == 100
    codes = _ffi.op_codes()
    assert codes["UZ_OP_CONV_FWD"] == 1 and len(set(codes.values())) == len(codes)


def test_phiseg_spec_equals_reference_state_dict():
    from unet_zoo_amd.models.phiseg import phiseg_spec
    for name in ("phiseg_small", "phiseg_full_digest"):
        _, meta = G.load(name)
        assert phiseg_spec(1, 2, meta["filters"]) == G.spec_of(meta)


def test_phiseg_structure_only_on_cpu_and_no_fallback(monkeypatch):
    from unet_zoo_amd.models.phiseg import PHISeg
    _, meta = G.load("phiseg_small")
    net = PHISeg(1, 2, meta["filters"], image_size=(1, 64, 64), device="cpu")
    assert list(net.state_dict().keys()) == [k for k, _, _ in G.spec_of(meta)]
    assert sum(p.numel() for p in net.parameters()) == net._ptab.n_params
    plan = net._build(2, 64, 64, True, True)
    cnt = plan.summary()
    # + the two head ops (zero the bound slots, measure the parameter bound); the 10 SampleZBlock tails (5 prior + 5 posterior levels) are
    # one op each instead of mu head + sigma head + sampling (Plan.latent_heads), their backward three ops instead of five
    assert cnt["fwd"] == 287 - 20 + 2 and cnt["bwd"] > 400
    codes = [o["code"] for o in plan.fwd_ops]
    assert codes.count("UZ_OP_LATENT_HEADS_FWD") == 10 and "UZ_OP_LATENT_FWD" not in codes
    bcodes = [o["code"] for o in plan.bwd_ops]
    assert bcodes.count("UZ_OP_LATENT_HEADS_BWD_DATA") == 10 and bcodes.count("UZ_OP_LATENT_HEADS_BWD_WEIGHT") == 10 and bcodes.count("UZ_OP_LATENT_BWD") == 10
    monkeypatch.setenv("UZ_FUSE_HEADS", "0")
    net0 = PHISeg(1, 2, meta["filters"], image_size=(1, 64, 64), device="cpu")
    plan0 = net0._build(2, 64, 64, True, True)
    assert plan0.summary()["fwd"] == 287 + 2 and len(plan0.bwd_ops) == len(plan.bwd_ops) + 20
    assert sorted(plan0.param_grads) == sorted(plan.param_grads)
    monkeypatch.delenv("UZ_FUSE_HEADS")
    unused = sorted(k for k in net._pmap if k not in plan.param_grads)
    assert len(unused) == 16 and all("upsampling_path.4" in k for k in unused)
    if not torch.cuda.is_available():
        with pytest.raises(_ffi.UzError):
            net.forward(torch.zeros(2, 1, 64, 64), torch.zeros(2, 1, 64, 64))


def test_full_phiseg_parameter_count():
    from unet_zoo_amd.models.phiseg import phiseg_spec
    spec = phiseg_spec(1, 2, [32, 64, 128, 192, 192, 192, 192])
    n = 0
    for _, shape, kind in spec:
        if kind in ("conv_w", "conv_b", "bn_w", "bn_b"):
            k = 1
            for s in shape:
                k *= s
            n += k
    assert n == 24513330 and len(spec) == 820


def test_unet_and_probunet_specs_equal_reference_state_dict():
    from unet_zoo_amd.models.unet import unet_spec, Unet
    from unet_zoo_amd.models.probabilistic_unet import probunet_spec, ProbabilisticUnet
    _, meta = G.load("unet_small")
    assert unet_spec(1, 2, meta["filters"]) == G.spec_of(meta)
    _, meta = G.load("probunet_small")
    assert probunet_spec(1, 2, meta["filters"], meta["latent_dim"], 3) == G.spec_of(meta)
    net = ProbabilisticUnet(1, 2, meta["filters"], latent_dim=meta["latent_dim"], no_convs_fcomb=3, device="cpu")
    plan = net._build(2, 128, 128, True, True)
    assert sorted(k for k in net._pmap if k not in plan.param_grads) == sorted(meta["steps"][0]["none_grads"])
    # BASELINE config 2 / 3 parameter counts
    assert Unet(1, 2, [32, 64, 128, 192], device="cpu")._ptab.n_params == 2260194
    big = probunet_spec(1, 2, [32, 64, 128, 192, 192, 192, 192], 6, 3)
    n = sum(int(torch.tensor(s).prod()) if len(s) else 1 for _, s, kd in big if kd in ("conv_w", "conv_b", "bn_w", "bn_b"))
    assert n == 17956988


def test_kernel_resource_table_matches_the_sources_and_the_budgets():
    """profiles/kernel_resources.json (tools/kernel_resources.py: hipcc's own register / scratch / LDS report per kernel instance) must
    have been generated from the committed kernel sources, and the instances the training step lives on must stay inside their
    budgets.  Round 5 shipped, for a few hours, a wrapper loop that took the 32-channel-tile convolution from 72 to 340 bytes of scratch
    and the 16 x 16-pixel one from three to two workgroups per CU (32->32 @ 128 x 128: 60 -> 117 us) - invisible in every
    same-library A/B of the step."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "unet-zoo_amd", "csrc")
    tab = json.load(open(os.path.join(root, "profiles", "kernel_resources.json")))
    headers = [os.path.join(csrc, "uz_common.h"), os.path.join(csrc, "split_f16.h"), os.path.join(root, "include", "uz_api.h")]
    for f in sorted(os.listdir(csrc)):
        if not f.endswith(".hip"):
            continue
        h = hashlib.sha1()
        for p_ in [os.path.join(csrc, f)] + headers:
            h.update(open(p_, "rb").read())
        assert f in tab and tab[f]["sha1"] == h.hexdigest(), f"{f} changed since profiles/kernel_resources.json was generated: run tools/kernel_resources.py and LOOK at the diff"
    k = tab["conv_split.hip"]["kernels"]
    budgets = {  # instance: (max VGPRs, max scratch bytes per lane)
        "conv_splitp_kernel_1_512_32": (128, 80), "conv_split_kernel_1_512_32": (128, 140), "conv_splitp_kernel_1_256_16": (168, 16),
        "conv_split_kernel_1_256_16": (168, 0), "conv_splitp_db_kernel_2_512_32": (192, 0), "conv_split_db_kernel_2_512_32": (200, 0),
        "conv_splitp_bn_db_kernel_2_512_32": (200, 0), "conv_b16_db_kernel_4_512_32": (248, 0), "conv_b16_db_kernel_2_512_32": (168, 0)}
    for name, (vmax, smax) in budgets.items():
        assert k[name]["vgpr"] <= vmax and k[name]["scratch"] <= smax, (name, k[name])
    w = tab["conv_wgrad_split.hip"]["kernels"]
    heavy = [v for n, v in w.items() if n.startswith("wgrad_split_kernel<32, 64, 2") and n.endswith(", 0>")]      # the product's instances (last argument: MFMA shape)
    assert len(heavy) == 4 and all(v["vgpr"] <= 256 and v["scratch"] == 0 for v in heavy), heavy      # (236 - 254 since the bound predicate is accumulated over every tile)
    # the optional forms of round 6 (off in the product, NOTES_r6 section 8): the two-image form must not spill, the LDS-free small-plane
    # convolution has to fit beside the headline convolution's two waves x 184 VGPRs per SIMD (one wave: <= 144 registers, no LDS)
    assert w["wgrad_db_kernel<1, 1>"]["vgpr"] <= 256 and w["wgrad_db_kernel<1, 1>"]["scratch"] == 0, w["wgrad_db_kernel<1, 1>"]
    assert w["wgrad_half_kernel<1, 1>"]["vgpr"] <= 256 and w["wgrad_half_kernel<1, 1>"]["scratch"] == 0, w["wgrad_half_kernel<1, 1>"]      # two per CU
    cm = tab["conv_mfma.hip"]["kernels"]
    for n in ("conv_free_kernel<false>", "conv_free_kernel<true>"):
        assert cm[n]["vgpr"] <= 128 and cm[n]["scratch"] == 0 and not cm[n]["lds"], (n, cm[n])
    # Footprint discipline of the deep levels' BatchNorm launches: the headline convolution holds two waves x 184 VGPRs per SIMD, so a
    # workgroup with W waves per SIMD starts beside it only under (512 - 2 * 184) / W registers (tools/bench_coresidency.py: 5 us alone,
    # 9 us beside the convolution under the budget, 65 - 77 us above it)
    # the chain launch (csrc/chain.hip): 1024-thread workgroups = four waves per SIMD need <= 128 registers, and its matrix loop carries no spill
    ck = tab["chain.hip"]["kernels"]
    assert all(v["vgpr"] <= 128 and v["scratch"] <= 400 and v["lds"] <= 32 * 1024 for n, v in ck.items() if n.startswith("chain_kernel")), ck
    free = 512 - 2 * k["conv_splitp_db_kernel_2_512_32"]["vgpr"]
    b = tab["bn.hip"]["kernels"]
    for name, waves in (("bn_fused_small_fwd<2>", 1), ("bn_fused_small_bwd<2>", 1), ("bn_fused_small_fwd<8>", 1), ("bn_fused_small_bwd<8>", 1),
                        ("bn_fused_mid_fwd<false, 512, 4>", 2), ("bn_fused_mid_fwd<true, 512, 4>", 2), ("bn_fused_mid_bwd<512, 4>", 2)):
        assert -(-b[name]["vgpr"] // 8) * 8 * waves <= free, (name, b[name], free)


def _check_lane_schedule(plan, which, ops):
    """Brute force: every pair of ops touching overlapping memory with at least one write must be
    ordered by lane order + event waits; groups stay contiguous on one lane (private scratch)."""
    sc, n = plan.scheds[which], len(ops)
    hb = [0] * n                                    # bitset of ops that happen before op k
    last = {}
    for k, o in enumerate(ops):
        m = 0
        if o["lane"] in last:
            j = last[o["lane"]]
            m |= hb[j] | (1 << j)
        for w in range(sc[k].n_wait):
            j = sc[k].wait[w]
            assert j < k and sc[j].signal and ops[j]["lane"] != o["lane"]
            m |= hb[j] | (1 << j)
        hb[k] = m
        last[o["lane"]] = k
        if k and ops[k - 1]["gid"] == o["gid"]:
            assert ops[k - 1]["lane"] == o["lane"]
    acc = [plan._access(o) for o in ops]            # (reads, writes, accumulates): accumulates (bound slots) commute with each other only

    def overlap(xs, ys):
        return any(a[0] == b[0] and a[1] < b[2] and b[1] < a[2] for a in xs for b in ys)
    pairs = amax_pairs = 0
    for j in range(n):
        rj, wj, aj = acc[j]
        for k in range(j + 1, n):
            rk, wk, ak = acc[k]
            slot = overlap(aj, rk) or overlap(rj, ak)           # a bound slot's accumulator against one of its readers
            if overlap(wj, rk) or overlap(wj, wk) or overlap(rj, wk) or overlap(wj, ak) or overlap(aj, wk) or slot:
                pairs += 1
                amax_pairs += bool(slot)
                assert (hb[k] >> j) & 1, (which, j, ops[j]["code"], k, ops[k]["code"])
    return pairs, len({o["lane"] for o in ops}), amax_pairs


@pytest.mark.parametrize("lanes", ["1", "3", "4"])
def test_lane_schedule_preserves_every_dependency(lanes, monkeypatch):
    from unet_zoo_amd.models.phiseg import PHISeg
    from unet_zoo_amd.models.probabilistic_unet import ProbabilisticUnet
    monkeypatch.setenv("UZ_LANES", lanes)
    _, meta = G.load("phiseg_small")
    net = PHISeg(1, 2, meta["filters"], image_size=(1, 64, 64), device="cpu")
    plan = net._build(2, 64, 64, True, True)
    for which, ops in (("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops)):
        pairs, used, amax_pairs = _check_lane_schedule(plan, which, ops)
        assert pairs > 100 and used == min(int(lanes), 4) and amax_pairs > 20
    _, meta = G.load("probunet_small")
    pu = ProbabilisticUnet(1, 2, meta["filters"], latent_dim=meta["latent_dim"], no_convs_fcomb=3, device="cpu")
    for plan in (pu._build(2, 64, 64, True, True), pu._build(2, 64, 64, False, False)):
        for which, ops in (("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops), *plan.extra_ops.items()):
            _check_lane_schedule(plan, which, ops)


@pytest.mark.parametrize("lanes", ["2", "3"])
def test_bound_slot_readers_are_ordered_against_every_accumulator_at_the_headline_size(lanes, monkeypatch):
    """a magnitude-bound slot is shared by all producers of a concat buffer; a consumer of a channel sub-range
    derives its operand scale from the slot several times while it runs, so NO producer of the slot may run beside it - also one
    that writes channels the consumer never reads.  Walks every (accumulator, reader) pair of every slot of the headline plan
    (PHiSeg 7 / 5, batch 32, 128 x 128) and asserts that the schedule orders it, in either direction."""
    from unet_zoo_amd.models.phiseg import PHISeg
    monkeypatch.setenv("UZ_LANES", lanes)
    net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], latent_levels=5, image_size=(1, 128, 128), device="cpu")
    plan = net._build(32, 128, 128, True, True)
    total = 0
    for which, ops in (("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops)):
        sc, n = plan.scheds[which], len(ops)
        hb, last = [0] * n, {}
        for k, o in enumerate(ops):
            m = 0
            if o["lane"] in last:
                m |= hb[last[o["lane"]]] | (1 << last[o["lane"]])
            for w in range(sc[k].n_wait):
                m |= hb[sc[k].wait[w]] | (1 << sc[k].wait[w])
            hb[k] = m
            last[o["lane"]] = k
        accs, reads = {}, {}
        for k, o in enumerate(ops):
            r_, w_, a_ = plan._access(o)
            for sp, lo, hi in a_:
                accs.setdefault(lo, []).append(k)
            for sp, lo, hi in r_:
                if sp == ("amax",):
                    reads.setdefault(lo, []).append(k)
        for slot, ks in accs.items():
            for a in ks:
                for r in reads.get(slot, ()):
                    if a == r:
                        continue
                    lo_, hi_ = min(a, r), max(a, r)
                    total += 1
                    assert (hb[hi_] >> lo_) & 1, (which, slot, a, ops[a]["code"], r, ops[r]["code"])
    assert total > 300


@pytest.mark.parametrize("cost_model", ["alone", "beside"])
def test_rescheduling_with_measured_costs_keeps_every_dependency_and_restores_the_first_schedule(cost_model, monkeypatch):
    """Engine.tune_schedule's CPU half: per-op measured durations (`cost_us`) replace the cost model, Plan.reschedule starts again from
    the PROGRAM order (not from the previous schedule's order), the result is a different but equally valid schedule, and taking the
    measured costs away again reproduces the first schedule op for op."""
    import random
    from unet_zoo_amd.models.phiseg import PHISeg
    monkeypatch.setenv("UZ_LANES", "3")
    monkeypatch.setenv("UZ_SCHED_HEAVY", "off")
    monkeypatch.setenv("UZ_SCHED_COST", cost_model)
    _, meta = G.load("phiseg_small")
    net = PHISeg(1, 2, meta["filters"], image_size=(1, 64, 64), device="cpu")
    plan = net._build(2, 64, 64, True, True)
    rng = random.Random(7)
    for which, ops in (("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops)):
        first = [(id(o), o["lane"]) for o in ops]
        members = {id(o) for o in ops}
        for o in ops:
            o["cost_us"] = rng.choice([3.0, 30.0, 300.0])
        plan.reschedule(which)
        assert {id(o) for o in ops} == members and len(ops) == len(first) == plan.tapes[which][1]
        assert [(id(o), o["lane"]) for o in ops] != first
        pairs, used, _ = _check_lane_schedule(plan, which, ops)
        assert pairs > 100 and used == 3
        for o in ops:
            del o["cost_us"]
        plan.reschedule(which)
        assert [(id(o), o["lane"]) for o in ops] == first
        _check_lane_schedule(plan, which, ops)


# ----------------------------------------------------------------------------- U2: the three initialisers
def _moments(t):
    a = t.detach().double()
    return dict(n=a.numel(), mean=float(a.mean()), std=float(a.std(unbiased=False)), absmax=float(a.abs().max()))


@pytest.mark.parametrize("which", ["unet", "phiseg", "probunet"])
def test_initialisers_match_reference_moments(which):
    """utils.init_weights (kaiming-normal fan-in + truncated-normal bias, utils.py:69-83),
    init_weights_orthogonal_normal (utils.py:86-90, Fcomb), kaiming-normal + N(0,1) bias of
    AxisAlignedConvGaussian.conv_layer (probabilistic_unet.py:99-100) and torch's default Conv2d / BatchNorm2d init
    that PHiSeg keeps (phiseg.py:36 commented out).  Initial weights are random, so parity is distributional: every
    tensor's moments must agree with those of the REAL reference's freshly constructed model (tests/golden/
    init_moments.json, tools/gen_golden.py `init`) within sampling error, constants must be exact, Fcomb weights
    orthogonal, and truncated biases inside +-2 std."""
    import math
    from unet_zoo_amd.models import PHISeg, ProbabilisticUnet, Unet
    with open(os.path.join(G.GOLDEN, "init_moments.json")) as f:
        ref = json.load(f)[which]
    torch.manual_seed(123)
    nf7 = [32, 64, 128, 192, 192, 192, 192]
    net = {"unet": lambda: Unet(1, 2, [32, 64, 128, 192], device="cpu"),
           "phiseg": lambda: PHISeg(1, 2, nf7, latent_levels=5, image_size=(1, 128, 128), device="cpu"),
           "probunet": lambda: ProbabilisticUnet(1, 2, nf7, latent_dim=6, no_convs_fcomb=3, image_size=(1, 128, 128), device="cpu")}[which]()
    sd = {k: v for k, v in net.state_dict().items() if v.dtype.is_floating_point}
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        r, m = ref[k], _moments(v)
        assert m["n"] == r["n"], k
        if r["std"] == 0.0:                                   # BN affine / running statistics: exact constants
            assert m["std"] == 0.0 and m["mean"] == r["mean"], k
            continue
        n = r["n"]
        if n >= 256:
            # std of a sample of n iid values has relative sampling error ~ 1/sqrt(2n) (x ~1.3 for the uniform / truncated shapes)
            assert abs(m["std"] / r["std"] - 1.0) <= 6.0 / math.sqrt(2 * n) + 1e-3, (k, m, r)
            assert abs(m["mean"] - r["mean"]) <= 6.0 * r["std"] * math.sqrt(2.0 / n), (k, m, r)
        else:                                                 # tiny tensors (biases of 2..192 entries): same scale
            assert 0.3 <= m["std"] / r["std"] <= 3.0 if n >= 12 else m["absmax"] <= 10 * max(r["absmax"], r["std"]), (k, m, r)
        # bounded distributions keep their bound: uniform(+-b) and the truncated normal (+-2 std = 2e-3)
        if k.endswith(".bias") and r["absmax"] <= 2e-3 and "conv_layer" not in k and "last_conv" not in k and which != "phiseg":
            assert m["absmax"] <= 2e-3 + 1e-9, (k, m)
        if which == "phiseg" and v.dim() in (1, 4) and "convolution.1" not in k:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3] if v.dim() == 4 else None
            if fan_in:
                assert m["absmax"] <= 1.0 / math.sqrt(fan_in) + 1e-9, (k, m)      # kaiming_uniform(a=sqrt(5)) bound
        if "orth_err" in r:
            w = v.detach().double().reshape(v.shape[0], -1)
            gm = w @ w.t() if w.shape[0] <= w.shape[1] else w.t() @ w
            assert float((gm - torch.eye(gm.shape[0], dtype=torch.float64)).abs().max()) <= 1e-5, k


def test_unet_plan_folds_the_relu_backward_into_the_last_writer_of_dA(monkeypatch):
    """Plan-level view of the folded ReLU backward: in the vanilla U-Net at BASELINE size every Conv -> ReLU unit
    whose dA is last written by a split-path data gradient, a pooling or an interpolation backward loses its uz_relu_bwd op; the bias
    gradients come from ONE table-driven launch - or from one launch per unit under data parallelism, where a bucket's gradients must
    be final when its all-reduce starts; UZ_FOLD_RELU_BWD=0 restores the plain tape.  The lane schedule stays dependency-correct."""
    from unet_zoo_amd.models.unet import Unet
    count = lambda plan, code: sum(o["code"] == code for o in plan.bwd_ops)
    net = Unet(1, 2, [32, 64, 128, 192], device="cpu")
    plan = net._build(32, 128, 128)
    n_units = sum(o["code"] == "UZ_OP_CONV_FWD" and o["i"][8] == 1 for o in plan.fwd_ops)          # convolutions with the fused forward ReLU
    assert n_units == 21
    assert count(plan, "UZ_OP_RELU_BWD") == 1 and count(plan, "UZ_OP_CHAN_SUM_TABLE") == 1 and count(plan, "UZ_OP_CHAN_SUM_PARTIALS") == 0
    folded = [o for o in plan.bwd_ops if o["code"] in ("UZ_OP_CONV_BWD_DATA", "UZ_OP_AVGPOOL_BWD", "UZ_OP_BILINEAR_BWD") and len(o["p"]) > 7 - 5 * (o["code"] != "UZ_OP_CONV_BWD_DATA")]
    assert len(folded) == 20
    monkeypatch.setenv("UZ_LANES", "2")
    plan2 = Unet(1, 2, [32, 64, 128, 192], device="cpu")._build(32, 128, 128)
    _check_lane_schedule(plan2, "bwd", plan2.bwd_ops)
    monkeypatch.setenv("UZ_FOLD_RELU_BWD", "0")
    plain = Unet(1, 2, [32, 64, 128, 192], device="cpu")._build(32, 128, 128)
    assert count(plain, "UZ_OP_RELU_BWD") == 21 and count(plain, "UZ_OP_CHAN_SUM_TABLE") == 0


def test_phiseg_plan_round4_passes(monkeypatch):
    """Plan-level view of split storage at the BASELINE size (Plan._round4_passes): every buffer kept as operand
    pieces is written only by launches that know their bound beforehand (BatchNorm apply with the convolution's statistics, pooling,
    interpolation) and read only by split-path convolutions; a concat buffer's two producers own one bound slot each and the
    forward consumer switches scales on a 16-channel boundary; dy is packed exactly where the unit's backward is not the
    one-launch kernel; the deferred conv-bias sums are one table-driven launch; the switches restore the plain tape."""
    from unet_zoo_amd import _ffi
    from unet_zoo_amd._plan import View
    from unet_zoo_amd.models.phiseg import PHISeg
    L = _ffi.lib()
    if L.uz_get_conv_math() in (0, 3):
        pytest.skip("two-piece operands only exist in the default / split math modes")
    net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], device="cpu")
    net.train()
    plan = net._build(32, 128, 128, True, True)
    info = plan.round4
    assert info["act_packed"] >= 30 and info["folded"] == 0
    packed = [b for b in plan.bufs if b.packed]
    assert len(packed) == info["act_packed"]
    named = {id(v.buf) for k, v in plan.io.items() if isinstance(v, View)}
    all_ops = plan.fwd_ops + plan.loss_ops + plan.bwd_ops
    for b in packed:
        assert id(b) not in named
        slots = set()
        for o in all_ops:
            for j, r in enumerate(o["p"]):
                if not (isinstance(r, View) and r.buf is b):
                    continue
                c, i = o["code"], o["i"]
                if c == "UZ_OP_BN_RELU_FWD":
                    assert j == 6 and i[10] == 1 and o["p"][8] is not None
                    assert i[8] > 0 or 4096 < i[3] * i[4] * i[5] <= L.uz_bn_fwd_fused_limit(i[4], i[5])      # statistics from the convolution's partials, or the one-launch mid path
                    slots.add(o["p"][8][:2])       # (producers tag the ref "acc": compare slot numbers)
                elif c in ("UZ_OP_BILINEAR_FWD", "UZ_OP_AVGPOOL_FWD"):
                    assert j == 1 and i[7 if c == "UZ_OP_BILINEAR_FWD" else 6] == 1 and o["p"][2] is not None
                    slots.add(o["p"][3][:2])
                elif c == "UZ_OP_CONV_FWD":
                    assert j == 0 and i[10] == 1 and L.uz_conv_route(0, i[0], i[2], i[4], i[5], i[6], 3) == 1
                    assert i[11] % 16 == 0 and (i[11] == 0) == (o["p"][10] is None)
                    assert o["p"][5] in slots and (o["p"][10] is None or o["p"][10] in slots)
                elif c == "UZ_OP_CONV_BWD_WEIGHT":
                    assert j == 0 and i[8] == 1 and L.uz_conv_route(2, i[0], i[2], i[4], i[5], i[6], 3) == 1
                else:
                    raise AssertionError(f"{c} touches the packed buffer {b.name}")
    # the heaviest layer (3 x 3, 224 -> 128 @ 128 x 128) reads its concat input as two segments: 32 channels from the BatchNorm apply, 192 from the interpolation
    big = [o for o in plan.fwd_ops if o["code"] == "UZ_OP_CONV_FWD" and o["i"][:3] == [224, 224, 128]]
    assert len(big) == 1 and big[0]["i"][10:12] == [1, 32]
    # dy: packed where the backward is the three-launch large-plane path, fp32 where it is one launch with the channel's batch on chip
    for o in plan.bwd_ops:
        if o["code"] == "UZ_OP_BN_RELU_BWD":
            i = o["i"]
            npx, fused = i[4] * i[5] * i[6], L.uz_bn_bwd_fused_limit(i[5], i[6])
            assert not (i[9] and npx <= fused)                        # dy is packed only on the three-launch path (bound known before the apply pass)
            assert (i[10] == 1) == (npx > fused)                      # conv-bias partial rows only on the three-launch path
    n3 = sum(o["code"] == "UZ_OP_BN_RELU_BWD" and o["i"][10] == 1 for o in plan.bwd_ops)
    assert 10 <= info["dy_packed"] <= n3
    assert sum(o["code"] == "UZ_OP_CHAN_SUM_TABLE" for o in plan.bwd_ops) == (1 if n3 else 0)
    # weight-gradient slab reductions: one table-driven launch for all 3 x 3 layers, none inside the layers' own ops
    tab = [o for o in plan.bwd_ops if o["code"] == "UZ_OP_WGRAD_REDUCE_TABLE"]
    wg = [o for o in plan.bwd_ops if o["code"] == "UZ_OP_CONV_BWD_WEIGHT" and o["i"][7] == 3]
    assert len(tab) == 1 and tab[0]["i"][0] == len(wg) == 106 and all(o["i"][11] == 1 and o["p"][8] is not None for o in wg)
    monkeypatch.setenv("UZ_LANES", "2")
    plan2 = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], device="cpu")
    plan2.train()
    plan2 = plan2._build(32, 128, 128, True, True)
    _check_lane_schedule(plan2, "bwd", plan2.bwd_ops)
    _check_lane_schedule(plan2, "fwd", plan2.fwd_ops)
    # small planes: the split-K data gradients whose only reader is a unit's BatchNorm backward leave slabs for it (no reduce launch)
    dg = [o for o in plan.bwd_ops if o["code"] == "UZ_OP_CONV_BWD_DATA" and len(o["i"]) > 10 and o["i"][10] == 3]
    assert len(dg) == info["dgrad_folded"] >= 20
    for o in dg:
        i = o["i"]
        assert i[4] * i[5] * i[6] <= 4096 and L.uz_conv_bwd_splitk_parts(i[2], i[0], i[4], i[5], i[6], i[7]) > 1
        users = [b for b in plan.bwd_ops if b["code"] == "UZ_OP_BN_RELU_BWD" and len(b["p"]) > 11 and b["p"][11] is o["p"][7]]
        assert len(users) == 1 and users[0]["i"][11] == L.uz_conv_bwd_splitk_parts(i[2], i[0], i[4], i[5], i[6], i[7])
        assert plan.bwd_ops.index(users[0]) > plan.bwd_ops.index(o)
    for k in ("UZ_PACK_ACT", "UZ_PACK_DY", "UZ_DBIAS_TABLE", "UZ_WGRAD_TABLE", "UZ_BN_FOLD_DGRAD"):
        monkeypatch.setenv(k, "0")
    net3 = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], device="cpu")
    net3.train()
    plain = net3._build(32, 128, 128, True, True)
    assert plain.round4 == dict(folded=0, dy_packed=0, act_packed=0, act_views=0, dgrad_folded=0) and not any(b.packed for b in plain.bufs)
    assert sum(o["code"] in ("UZ_OP_CHAN_SUM_TABLE", "UZ_OP_WGRAD_REDUCE_TABLE") for o in plain.bwd_ops) == 0


def test_default_routing_keeps_small_planes_off_the_split_path():
    """The split path's error is bounded relative to the TENSOR's maximum (2^-22 |a|max |b|), not per element: on tiny planes
    whose gradients mix channels of very different magnitude that is coarser than fp32's per-element rounding (the one gradient
    gate that fails with UZ_CONV_MATH=split forced everywhere sits on the 8 x 8 ... 2 x 2 levels of a batch-2 fixture - VERDICT r3
    item 6c).  The DEFAULT policy therefore never routes a plane below 16 x 16 to it, whatever the channel counts or the batch:
    uz_conv_route is the single place that decides, and this pins it."""
    L = _ffi.lib()
    mode = L.uz_get_conv_math()
    try:
        assert L.uz_set_conv_math(1) == 0
        for kind in (0, 1, 2):
            for h, w in ((2, 2), (4, 4), (8, 8), (8, 16), (16, 8), (12, 12), (15, 64)):
                for cin, cout in ((32, 32), (64, 64), (192, 192), (256, 256), (768, 256), (2, 192), (192, 2)):
                    for n in (1, 2, 32, 256, 4096):
                        assert L.uz_conv_route(kind, cin, cout, n, h, w, 3) != 1, (kind, cin, cout, n, h, w)
        # ... and 1 x 1 kernels never take it
        assert all(L.uz_conv_route(k, 192, 192, 32, 64, 64, 1) != 1 for k in (0, 1, 2))
        # the large planes of the BASELINE configs do
        assert all(L.uz_conv_route(k, 224, 128, 32, 128, 128, 3) == 1 for k in (0, 1, 2))
        assert L.uz_set_conv_math(0) == 0 and all(L.uz_conv_route(k, 224, 128, 32, 128, 128, 3) == 0 for k in (0, 1, 2))
    finally:
        L.uz_set_conv_math(-1 if os.environ.get("UZ_CONV_MATH") is None else mode)


def test_bench_refuses_work_skipping_knobs_and_reports_what_the_binary_is():
    """the line must prove it ran the product.  bench.py exits with status 2 when UZ_DIAG_SKIP or UZ_LIB is set (or
    the library says it is an experiment build) unless --allow-experiment is given, and then marks the line; config.build carries
    uz_build_info() (three piece products, no experiment / diagnostic code, the source hash), config.env every UZ_* variable; the
    product library has no diag_skip code at all (it is behind -DUZ_DIAG)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "UZ_LIB", "UZ_DIAG_SKIP")}
    run = lambda env, *flags: subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", *flags], env=dict(base, UZ_BENCH_DRY="1", **env),
                                             capture_output=True, text=True, timeout=300)
    ok = run({})
    assert ok.returncode == 0, ok.stderr[-1000:]
    d = json.loads(ok.stdout.strip().splitlines()[-1])
    b = d["config"]["build"]
    assert b["experiment"] == 0 and b["products_per_mac"] == 3 and b["diag_skip_compiled"] == 0 and b["variant"] == "" and len(b["source_hash"]) == 16
    assert d["config"]["env"].get("UZ_BENCH_DRY") == "1" and "experiment" not in d
    for env in (dict(UZ_DIAG_SKIP="conv:8"), dict(UZ_LIB=os.path.join(root, "unet-zoo_amd", "libuz_hip.so"))):
        r = run(env)
        assert r.returncode == 2 and not r.stdout.strip() and "refusing" in r.stderr and list(env)[0] in r.stderr, (env, r.returncode, r.stderr[-500:])
        a = run(env, "--allow-experiment")
        assert a.returncode == 0 and json.loads(a.stdout.strip().splitlines()[-1])["experiment"] is True
    syms = subprocess.run(["nm", "-C", os.path.join(root, "unet-zoo_amd", "libuz_hip.so")], capture_output=True, text=True).stdout
    assert "diag_skip" not in syms and "uz_build_info" in syms


def test_bench_cu_share_of_a_reduced_grid_weight_gradient():
    """bench.py cu_share: the share of the chip a tape op's launch can occupy - 1 for everything except the split-path weight gradient, whose
    grid is (channel tiles) x (slabs); with the PHiSeg target of 128 workgroups the heaviest layer's launch covers half the CUs (the family's
    chip_ms_per_step / frac_on_occupied_cus in the bench line are built from it)."""
    import importlib.util
    from unet_zoo_amd import _ffi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("uz_bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    L = _ffi.lib()
    if L.uz_get_conv_math() in (0, 3):
        pytest.skip("split-path weight gradient only")
    old = L.uz_get_wgrad_target()
    try:
        L.uz_set_wgrad_target(128)
        wg = dict(code="UZ_OP_CONV_BWD_WEIGHT", i=[224, 224, 128, 128, 32, 128, 128, 3])
        assert bench.cu_share(wg, L) == 0.5
        L.uz_set_wgrad_target(256)
        assert bench.cu_share(wg, L) == 1.0
        assert bench.cu_share(dict(code="UZ_OP_CONV_FWD", i=[224, 224, 128, 128, 32, 128, 128, 3]), L) == 1.0
        assert bench.cu_share(dict(code="UZ_OP_CONV_BWD_WEIGHT", i=[192, 192, 192, 192, 32, 4, 4, 3]), L) == 1.0       # fp32 kernels: full grid
    finally:
        L.uz_set_wgrad_target(old if old != 256 else 0)


def _happens_before(plan, which, ops):
    sc, n = plan.scheds[which], len(ops)
    hb, last = [0] * n, {}
    for k, o in enumerate(ops):
        m = 0
        if o["lane"] in last:
            m |= hb[last[o["lane"]]] | (1 << last[o["lane"]])
        for w in range(sc[k].n_wait):
            m |= hb[sc[k].wait[w]] | (1 << sc[k].wait[w])
        hb[k] = m
        last[o["lane"]] = k
    return hb


def test_chain_pass_builds_convex_levelled_chains_at_the_headline_size(monkeypatch):
    """Plan._chain_pass (csrc/chain.hip, UZ_CHAIN=8192: off by default - DESIGN.md section 9): the forward tape's 16 x 16 ... 2 x 2 ops become ONE
    chain op, the backward tape's one per sub-network.  Checked on the headline plan: (a) inside a chain a sub-op sits in a LATER phase
    than every sub-op it depends on (the hazard analysis of the per-op tape, restricted to the set); (b) every conflicting pair of the
    rewritten tape - the chain op carries the union of its sub-ops' accesses - stays ordered by the lane schedule; (c) no scheduling
    group that shares lane scratch is cut in two by a chain (a unit's BatchNorm backward and its weight gradient stay back to back);
    (d) the tiles of every sub-op are counted by the library and the chains' phase tables are consistent."""
    from unet_zoo_amd import _ffi
    from unet_zoo_amd.models.phiseg import PHISeg
    if _ffi.lib().uz_get_conv_math() in (0, 3):
        pytest.skip("the chain's convolutions are two-piece split-fp16")
    monkeypatch.setenv("UZ_CHAIN", "8192")
    net = PHISeg(1, 2, [32, 64, 128, 192, 192, 192, 192], device="cpu")
    net.train()
    plan = net._build(32, 128, 128, True, True)
    info = plan.chain_info
    assert len(info["fwd"]) == 1 and info["fwd"][0]["ops"] > 140 and [c["net"] for c in info["bwd"]] == ["likelihood", "prior", "posterior"]
    assert sum(o["code"] == "UZ_OP_CHAIN" for o in plan.fwd_ops) == 1 and sum(o["code"] == "UZ_OP_CHAIN" for o in plan.bwd_ops) == 3
    assert sum(o["code"] == "UZ_OP_CHAIN_PACK" for o in plan.fwd_ops) == 1 and sum(o["code"] == "UZ_OP_CHAIN_PACK" for o in plan.bwd_ops) == 1
    small = lambda o: o["code"] in ("UZ_OP_CONV_FWD", "UZ_OP_CONV_BWD_DATA") and o["i"][7] == 3 and o["i"][4] * o["i"][5] * o["i"][6] <= 8192
    assert not any(small(o) for o in plan.fwd_ops + plan.bwd_ops)            # every small 3 x 3 forward / data gradient went into a chain
    for ch in plan._chains:
        sub = ch["sub"]
        origs = [e["orig"] for e in sub]
        # (a) phases respect the dependencies of the original ops
        deps = plan._hazard_deps(sorted({id(o): o for o in origs}.values(), key=lambda o: min(e["k"] for e in sub if e["orig"] is o)))
        order = sorted({id(o): o for o in origs}.values(), key=lambda o: min(e["k"] for e in sub if e["orig"] is o))
        first = {id(o): min(e["level"] for e in sub if e["orig"] is o) for o in order}
        last = {id(o): max(e["level"] for e in sub if e["orig"] is o) for o in order}
        for k, o in enumerate(order):
            for d in deps[k]:
                assert last[id(order[d])] < first[id(o)], (ch["net"], order[d]["code"], o["code"])
        # (d) phase table
        dev = plan._chain_tables(plan._chains.index(ch))
        ph = dev["phases"].reshape(-1, 2).tolist()
        assert sum(n for _, n in ph) == len(sub) and [a for a, _ in ph] == sorted(a for a, _ in ph) and dev["n_phases"] == 1 + max(e["level"] for e in sub)
        assert all(e["ntiles"] > 0 and 0 <= e["tile0"] < ch["n_wgs"] for e in sub)
    for which, ops in (("fwd", plan.fwd_ops), ("bwd", plan.bwd_ops)):
        # (b) every conflicting pair stays ordered
        pairs, used, _ = _check_lane_schedule(plan, which, ops)
        assert pairs > 100
        # (c) a unit whose dy lives in the lane scratch keeps BatchNorm backward, weight gradient and data gradient back to back
        for k, o in enumerate(ops):
            if o["code"] == "UZ_OP_BN_RELU_BWD" and type(o["p"][5]).__name__ == "_ScratchView" and o["p"][5].view is None:
                mates = [q for q in ops if q.get("gid") == o["gid"]]
                pos = [ops.index(q) for q in mates]
                assert pos == list(range(pos[0], pos[0] + len(pos))) and len({q["lane"] for q in mates}) == 1, (which, k)
