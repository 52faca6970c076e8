"""Common runtime of the native models: flat parameter storage, plan cache, tape execution,
the autograd bridge that makes ``loss.backward()`` run the backward tape, and the data-parallel
gradient all-reduce hook.

Product path only: every pass runs hand-written HIP kernels through ``libuz_hip.so``.  A model can
be *constructed* without a GPU (state_dict surface, plan building - used by the CPU test tier) but
any attempt to execute raises: there is no CPU fallback.
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from . import _ffi
from ._modtree import attach
from ._plan import ParamTable, Plan


def set_conv_math(mode):
    """uz_set_conv_math (0 fp32 MFMA | 1 default | 2 split everywhere | 3 bf16 | -1 back to UZ_CONV_MATH).  The mode is
    process-global and every plan of every live model was built for the mode in force at the time (routing, workspace and
    packed-image sizes, folded ops): NativeModel._plan compares the library's current mode with the one its plans were built
    under and drops plans and graphs when they differ - whoever flipped the switch."""
    _ffi.check(_ffi.lib().uz_set_conv_math(int(mode)), "set_conv_math")


def default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class _TapeLoss(torch.autograd.Function):
    """loss = run(loss tape); backward = run(backward tape) and publish parameter gradients."""

    @staticmethod
    def forward(ctx, anchor, model, plan):
        ctx.model, ctx.plan = model, plan
        plan.run("loss", model._stream())
        return plan.tensor(plan.total).reshape(()).clone()

    @staticmethod
    def backward(ctx, gout):
        ctx.model._run_backward(ctx.plan, gout)
        return None, None, None


class NativeModel(nn.Module):
    """Base class: owns the ParamTable and the per-shape plans."""

    def __init__(self):
        super().__init__()
        self._plans = {}
        self._cur = None
        self._dp_group = None
        self._use_graphs = False
        self._graphs = {}

    # ------------------------------------------------------------------ storage
    def _init_storage(self, spec, device=None):
        dev = torch.device(device) if device is not None else default_device()
        object.__setattr__(self, "_ptab", ParamTable(spec, dev))
        attach(self, self._ptab)
        object.__setattr__(self, "_anchor", torch.zeros(1, device=dev, requires_grad=True))
        self._pmap = dict(self.named_parameters())

    @property
    def device(self):
        return self._ptab.device

    def _apply(self, fn, recurse=True):
        # nn.Module.to()/cuda()/float(): parameters are views of one flat device buffer that the
        # kernels address directly, so they may not be moved or cast; same-device calls are no-ops.
        probe = fn(torch.zeros(1, device=self._ptab.device))
        if probe.device != self._ptab.device or probe.dtype != torch.float32:
            raise RuntimeError("native models are bound to their fp32 device buffers; construct the model on the target GPU")
        return self

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def check_bounds(self, clear=True):
        """Device flag word of the split-fp16 convolution path (uz_device_flags): 0 when every tensor stayed within the
        magnitude bound its consumer was given; bit 1 / 2 / 4 = an activation / weight / gradient exceeded its bound by more
        than 4x and was clamped (results wrong but finite).  The check is SAMPLED - every workgroup of a split kernel tests the first
        chunk / pixel tile it stages (a stale or wrong bound is a property of the whole tensor and shows there; the clamp itself
        covers every element), operands that arrive as split storage are not checked at all (their producer derived the scale from
        the bound it wrote them with).  Synchronises the stream - call it per epoch, not per step."""
        self._require_gpu()
        out = C.c_int(0)
        _ffi.check(_ffi.lib().uz_device_flags(C.byref(out), 1 if clear else 0, C.c_void_p(self._stream())), "device_flags")
        return out.value

    def guard_bounds(self):
        """check_bounds() + the safety net behind it: if any split-fp16 kernel saw a tensor beyond its magnitude bound (values were
        clamped: finite, but wrong), warn, switch the PROCESS to the fp32-MFMA kernels (set_conv_math(0): no bounds, no scales;
        every live model drops its plans and graphs).  Under data parallelism the decision is COLLECTIVE - the flag words are
        OR-ed over the ranks on the host control plane, so either every rank falls back and repeats the step or none does (a
        rank acting alone would issue a second set of all-reduces that pair with its peers' next step).  Returns the flag word;
        a caller that gets non-zero repeats the step it just ran from the state snapshot_step() saved (train_model.py does).
        Costs one stream synchronisation: meant for loops that synchronise per step anyway (the reference's scheduler does)."""
        flags = self.check_bounds()
        if getattr(self, "_dp", None) is not None:
            from . import dp
            flags = dp.or_flags(flags, self._dp.group)          # ONE collective for the three bits
        if flags:
            import warnings
            warnings.warn(f"split-fp16 convolution path: magnitude bound exceeded (flags {flags:#x}: 1 activation, 2 weight, 4 gradient) - "
                          "falling back to fp32 MFMA arithmetic for the rest of this process", RuntimeWarning)
            set_conv_math(0)
            self._plans.clear()              # (other live models notice the new mode in their next _plan() lookup)
            self._drop_graphs()
        return flags

    def snapshot_step(self):
        """State a repeated step must start from: BatchNorm running statistics, batch counters and the position of the
        latent-noise stream (a few KB, device-to-device, no synchronisation)."""
        snap = (self._ptab.bflat.clone(), self._ptab.nbt.clone(), self._rng_state().clone())
        object.__setattr__(self, "_step_snapshot", snap)

    def restore_step(self):
        """Undo what the forward pass since snapshot_step() did to the BatchNorm buffers / counters and rewind the noise stream,
        so that the repeated step draws the SAME eps and updates the running statistics once."""
        snap = self.__dict__.get("_step_snapshot")
        if snap is None:
            raise RuntimeError("restore_step() without snapshot_step()")
        self._ptab.bflat.copy_(snap[0])
        self._ptab.nbt.copy_(snap[1])
        self._rng_state().copy_(snap[2])

    def _require_gpu(self):
        if self.device.type != "cuda":
            raise _ffi.UzError("no GPU visible: the native path has no CPU fallback (model was built in structure-only mode)")

    # ------------------------------------------------------------------ plans
    def _plan(self, key, builder):
        mode = _ffi.lib().uz_get_conv_math()
        if self._plans and self.__dict__.get("_plans_mode", mode) != mode:      # the process switched its convolution math mode
            self._plans.clear()
            self._drop_graphs()
            self._cur = None
        self.__dict__["_plans_mode"] = mode
        p = self._plans.get(key)
        if p is None:
            p = builder()
            self._plans[key] = p
        return p

    default_lanes = 2          # dependency lanes of the captured graphs (UZ_LANES overrides); see ProbabilisticUnet
    # How enable_graphs() replays a tape: "lanes" (default since round 5) = the tape's DAG issued by the host on one HIP stream per
    # lane with one event per cross-lane edge (uz_run_tape_lanes); "graph" = one hipGraph per tape (the DAG captured by
    # uz_graph_create_lanes).  UZ_REPLAY overrides.  Measured on MI355X, same box: the ROCm 7.2 graph executor enqueues a captured DAG in
    # an order of its own and ready nodes wait behind unrelated ones in the in-order hardware queues (the prior's encoder of a PHiSeg
    # step started 1.3 ms late); host-issued lanes start every op when ITS predecessors are done: PHiSeg 17.2 -> 16.3 ms with three
    # lanes, Probabilistic U-Net 3 107 -> 3 240 images/s, U-Net / PHiSeg3D unchanged.  (More than four concurrently active hardware
    # queues collapse the step to 27 ms: lanes <= 4, GPU_MAX_HW_QUEUES >= lanes.)
    replay_mode = os.environ.get("UZ_REPLAY", "lanes")
    if replay_mode not in ("lanes", "graph"):          # (any other value used to select the hipGraph path silently: ADVICE r5)
        raise ValueError(f"UZ_REPLAY={replay_mode!r}: the replay mode is 'lanes' (host-issued lane replay) or 'graph' (hipGraph)")
    default_lanes_by_mode = {}  # per-model override of default_lanes for a replay mode, e.g. {"lanes": 3}
    # Planes (N*H*W pixels) up to which a layer's weight gradient becomes a scheduling group of its own (Plan._decouple_wgrad;
    # UZ_DECOUPLE_WGRAD overrides).  Measured A/B on one MI355X: PHiSeg 19.49 -> 19.26 ms with 8192 (the deep levels' backward
    # chains no longer carry the weight gradients), Probabilistic U-Net 11.33 -> 11.66 ms (three lanes already interleave its
    # chains) - so it is a per-model default.
    decouple_wgrad_px = 0
    # Planes (N*H*W pixels) up to which a tape's small-plane ops run as phases of persistent "chain" launches (Plan._chain_pass,
    # csrc/chain.hip; UZ_CHAIN overrides; 8192 = PHiSeg's 16 x 16 ... 2 x 2 levels at batch 32).  OFF by default: built, parity-green
    # and MEASURED SLOWER than the per-op tape on MI355X (round 6: PHiSeg step 15.7 ms per-op; 17.7 ms with the forward chain, 22.3 ms
    # with forward + backward chains - profiles/NOTES_r6.md says where the time goes).
    chain_px = 0
    decouple_wgrad_prefixes = ()      # sub-networks (parameter-name prefixes) whose weight gradients are decoupled at every size

    # Workgroups of the split-path weight gradients (uz_set_wgrad_target): a process setting of the library that sizes slab buffers at plan
    # time and grids at launch time - set in front of every plan build and every tape of THIS model, so models with different values can
    # alternate in one process.  256 = one workgroup per CU; PHISeg 128, ProbabilisticUnet 192 (measured under the lane replay).
    wgrad_workgroups = 256

    def _set_library_mode(self):
        _ffi.lib().uz_set_wgrad_target(int(self.wgrad_workgroups))

    def _new_plan(self, N, bn_training):
        self._set_library_mode()
        plan = Plan(N, self._ptab, bn_training, self.device)
        if "UZ_LANES" not in os.environ:
            plan.n_lanes = self.default_lanes_by_mode.get(self.replay_mode, self.default_lanes)
        # the scheduler's "one device-filling group at a time" rule pays in the graph replay only
        plan.sched_heavy = os.environ.get("UZ_SCHED_HEAVY", "off" if self.replay_mode == "lanes" else "r4")
        # the lane replay is scheduled with durations as it realises them beside other lanes' kernels (Plan._op_cost_beside), and
        # tune_schedule() refines them by measurement; the graph replay keeps the isolated-launch model its rule was measured with
        plan.sched_cost = os.environ.get("UZ_SCHED_COST", "beside" if self.replay_mode == "lanes" else "alone")
        plan.decouple_wgrad_px = self.decouple_wgrad_px
        plan.chain_px = self.chain_px
        plan.decouple_wgrad_prefixes = tuple(self.decouple_wgrad_prefixes)
        dp = getattr(self, "_dp", None)
        if dp is not None and dp.overlap:
            plan.grad_buckets = list(dp.buckets)
        # Stream budget (measured, DESIGN.md section 2: five or more concurrently active hardware queues collapse a PHiSeg step from 16 to
        # 27 ms): lanes + the data-parallel communication stream <= 4, and the process needs a hardware queue per lane
        extra = 1 if (dp is not None and dp.overlap) else 0
        if plan.n_lanes + extra > 4:
            import warnings
            if "UZ_LANES" in os.environ:
                warnings.warn(f"UZ_LANES={plan.n_lanes} with {extra} communication stream(s): more than four concurrently active streams - "
                              "measured to collapse the step (16 -> 27 ms on MI355X)", RuntimeWarning)
            else:
                plan.n_lanes = 4 - extra
        try:
            hwq = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
        except ValueError:
            hwq = 4
        if self.replay_mode == "lanes" and plan.n_lanes > 1 and hwq < plan.n_lanes and not self.__dict__.get("_warned_hwq"):
            import warnings
            self.__dict__["_warned_hwq"] = True
            warnings.warn(f"GPU_MAX_HW_QUEUES={hwq} < {plan.n_lanes} dependency lanes: lanes share hardware queues and serialise "
                          "(measured: 2 queues 1 741 images/s against 2 035 with 3, PHiSeg)", RuntimeWarning)
        return plan

    def enable_graphs(self, flag=True):
        """Replay forward/backward tapes as captured hipGraphs (one graph launch instead of ~1000
        kernel launches).  Capture happens lazily on a side stream after one eager warm-up run, over
        `plan.n_lanes` dependency lanes (UZ_LANES, default 2 - measured best on MI355X) so that independent chains of the
        tape - posterior / prior encoders, the likelihood branches - overlap on the device."""
        self._use_graphs = bool(flag)

    def _segments(self, plan, which):
        """Op ranges of a tape that are captured as separate hipGraphs.  Normally one.  The data-parallel backward tape is
        cut behind every bucket-final marker (UZ_OP_EVENT_RECORD): the marker itself is then recorded by the host on the
        compute stream between two graph launches - an event-record node INSIDE a hipGraph makes the ROCm 7.2 graph executor
        fall off its fast path (measured: 39 ms instead of 21 ms per step)."""
        arr, n = plan.tapes[which]
        ops = {"fwd": plan.fwd_ops, "bwd": plan.bwd_ops, "loss": plan.loss_ops}.get(which) or plan.extra_ops.get(which, [])
        cuts = [k for k, o in enumerate(ops) if o["code"] == "UZ_OP_EVENT_RECORD"]
        segs, start = [], 0
        for k in cuts:
            segs.append((start, k, ops[k]["p"][0][1]))          # [start, k) then record event of bucket b
            start = k + 1
        if start < n:
            segs.append((start, n, None))
        return segs

    def _capture(self, plan, which, a, b):
        arr, n = plan.tapes[which]
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        handle = C.c_void_p()
        sub = (_ffi.uz_op * (b - a)).from_address(C.addressof(arr) + a * C.sizeof(_ffi.uz_op))
        if plan.n_lanes > 1:
            full = plan.scheds[which]
            sch = (_ffi.uz_sched * (b - a))()
            for k in range(a, b):
                e, src = sch[k - a], full[k]
                e.lane, e.signal = src.lane, src.signal
                w = [src.wait[j] - a for j in range(src.n_wait) if src.wait[j] >= a]     # earlier segments have completed
                e.n_wait = len(w)
                for j, v in enumerate(w):
                    e.wait[j] = v
            _ffi.check(plan.L.uz_graph_create_lanes(sub, sch, b - a, plan.n_lanes, C.c_void_p(side.cuda_stream), C.byref(handle)),
                       f"graph capture '{which}'")
        else:
            _ffi.check(plan.L.uz_graph_create(sub, b - a, C.c_void_p(side.cuda_stream), C.byref(handle)), f"graph capture '{which}'")
        torch.cuda.current_stream(self.device).wait_stream(side)
        return handle

    def _run(self, plan, which):
        self._set_library_mode()
        if not self._use_graphs or which == "loss":
            plan.run(which, self._stream())
            return
        if self.replay_mode == "lanes" and plan.n_lanes > 1 and which in plan.scheds:
            # the tape's DAG issued by the host on one stream per lane, one event per cross-lane edge (uz_run_tape_lanes)
            arr, n = plan.tapes[which]
            rec = self.__dict__.get("_tune_rec")
            armed = False
            if n and rec is not None:                      # tune_schedule(): one timing event behind every op of this call
                plan.L.uz_lane_trace(1, n, None, 0)
                armed = True
            try:
                if n:
                    _ffi.check(plan.L.uz_run_tape_lanes(arr, plan.scheds[which], n, plan.n_lanes, C.c_void_p(self._stream())), f"lane replay '{which}'")
                if armed:
                    buf = (C.c_float * n)()
                    got = plan.L.uz_lane_trace(0, 0, buf, n)
                    armed = False
                    assert got == n, (got, n)
                    rec.setdefault((id(plan), which), (plan, []))[1].append([buf[k] * 1e3 for k in range(n)])
            finally:
                if armed:                                  # a replay that raised must not leave the tracer armed for the rest of the process
                    plan.L.uz_lane_trace(0, 0, None, 0)
            return
        key = (id(plan), which)
        g = self._graphs.get(key)
        if g is None:
            warm = plan.__dict__.setdefault("_warm", set())
            if which not in warm:          # first call: eager (also sets kernel attributes)
                warm.add(which)
                plan.run(which, self._stream())
                return
            g = [(self._capture(plan, which, a, b) if b > a else None, ev) for a, b, ev in self._segments(plan, which)]
            self._graphs[key] = g
        st = C.c_void_p(self._stream())
        for handle, ev in g:
            if handle is not None:
                _ffi.check(plan.L.uz_graph_launch(handle, st), f"graph launch '{which}'")
            if ev is not None:
                _ffi.check(plan.L.uz_event_record(plan.events[ev], st), "event_record")

    def tune_schedule(self, step, rounds=3, samples=5, damping=0.5, validate=8):
        """Profile-guided lane schedule (lane replay only; the counterpart of cudnn.benchmark for the tape scheduler).  `step` is the
        caller's closure that runs one forward / loss / backward on this model.  Each round replays it `samples` times with a timing event
        behind every op, sets every op's cost to the duration the replay REALISED for it beside the other lanes' kernels (start = the later
        of its lane predecessor and the ops it waits for; damped against the cost the current schedule was built from) and schedules the
        tapes again.  Per tape the fastest schedule seen (median tape time of its samples) is the candidate; the candidates are then timed
        against the schedules the call started with over `validate` un-instrumented steps each (twice, alternating) and the faster set
        stays - so the call never leaves a slower schedule behind than it found.  Any schedule of the DAG gives the same bits
        (Plan.reschedule).  Returns {"tape_us": {tape: [median microseconds per round]}, "step_ms": {"initial": a, "tuned": b}, "kept": which}."""
        assert self._use_graphs and self.replay_mode == "lanes", "tune_schedule() needs enable_graphs() with the lane replay"
        import statistics, time
        from ._plan import Plan
        from . import dp as _dp_mod
        grp = getattr(self, "_dp_group", None)
        if getattr(self, "_dp", None) is None:
            def allmax(v):
                return [float(x) for x in v]
        else:                                             # data parallel: one schedule for all ranks (the bucket exchange order follows it)
            def allmax(v):
                return _dp_mod.max_over_ranks(v, None if grp is True else grp)

        def tape_ops(plan, which):
            return {"fwd": plan.fwd_ops, "bwd": plan.bwd_ops}.get(which) or plan.extra_ops.get(which)

        def apply(costs):
            for (pid, which), (used, plan) in costs.items():
                for o in tape_ops(plan, which):
                    c = used[id(o)]
                    if c is None:
                        o.pop("cost_us", None)
                    else:
                        o["cost_us"] = c
                plan.reschedule(which)
        hist, best, first = {}, {}, {}
        for r in range(rounds + 1):
            self.__dict__["_tune_rec"] = rec = {}
            try:
                for _ in range(samples):
                    step()
            finally:
                self.__dict__["_tune_rec"] = None
            if r == 0 and not any(plan.n_lanes > 1 for plan, _runs in rec.values()):
                # single-lane plans (U-Net) have nothing to re-schedule: do not spend (rounds + 1) * samples more steps - with their
                # data-parallel exchanges - to return "initial" 
                return dict(tape_us={}, step_ms={}, kept="initial")
            for (pid, which), (plan, runs) in rec.items():          # (same plans, same order on every rank: the closure ran the same calls)
                if plan.n_lanes <= 1:
                    continue
                ops, sc = tape_ops(plan, which), plan.scheds[which]
                wall = allmax([statistics.median(max(e) for e in runs)])[0]
                hist.setdefault(which, []).append(round(wall, 1))
                used = {id(o): o.get("cost_us") for o in ops}
                first.setdefault((pid, which), (used, plan))
                if (pid, which) not in best or wall < best[(pid, which)][0]:
                    best[(pid, which)] = (wall, used, plan)
                if r == rounds:
                    continue
                cmodel = Plan._op_cost_beside if os.environ.get("UZ_SCHED_COST", plan.__dict__.get("sched_cost", "alone")) == "beside" else Plan._op_cost
                last, dur = {}, [0.0] * len(ops)
                for e in runs:
                    last.clear()
                    for k, o in enumerate(ops):
                        pred = ([last[o["lane"]]] if o["lane"] in last else []) + [sc[k].wait[w] for w in range(sc[k].n_wait)]
                        dur[k] += (e[k] - max((e[j] for j in pred), default=0.0)) / len(runs)
                        last[o["lane"]] = k
                dur = allmax(dur)
                for k, o in enumerate(ops):
                    old = o["cost_us"] if "cost_us" in o else cmodel(o) * 1e6
                    o["cost_us"] = max(1.0, damping * old + (1.0 - damping) * dur[k])
                plan.reschedule(which)
        if not best:
            return dict(tape_us=hist, step_ms={}, kept="initial")
        sets = {"initial": first, "tuned": {k: (used, plan) for k, (_w, used, plan) in best.items()}}
        ms = {"initial": [], "tuned": []}
        for _ in range(2):
            for name in ("initial", "tuned"):
                apply(sets[name])
                step()
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                for _ in range(validate):
                    step()
                torch.cuda.synchronize(self.device)
                ms[name].append(1e3 * (time.perf_counter() - t0) / validate)
        ms = {name: allmax(v) for name, v in ms.items()}
        kept = "tuned" if min(ms["tuned"]) < min(ms["initial"]) else "initial"
        apply(sets[kept])
        return dict(tape_us=hist, step_ms={k: round(min(v), 3) for k, v in ms.items()}, kept=kept)

    # ------------------------------------------------------------------ backward
    def set_data_parallel(self, group=True, overlap=True, backend=None):
        """Average gradients over the ranks (one process per GPU) inside loss.backward(): the flat fp32 gradient buffer
        is all-reduced bucket by bucket (one bucket per sub-network) over RCCL, each bucket as soon as the backward tape
        has finished it, on a communication stream beside the remaining backward kernels (dp.GradSync).  `group`: a
        torch.distributed group or True for the default group; backend None picks RCCL through the C ABI when the
        process group runs on nccl, torch.distributed collectives otherwise (gloo test double)."""
        from . import dp
        self._dp_group = group
        if getattr(self, "_dp", None) is not None:
            self._dp.close()                   # communicator and communication stream of an earlier call (streams are a scarce resource: dp.GradSync.close)
        self._dp = dp.GradSync(self, None if group is True else group, backend=backend, overlap=overlap)
        self._plans.clear()                # plans built before this call carry no bucket events
        self._drop_graphs()

    def _drop_graphs(self):
        """Destroy the captured hipGraphExec handles (they hold device allocations) instead of just forgetting them."""
        L = _ffi.lib() if self._graphs else None
        for segs in self._graphs.values():
            for handle, _ev in segs:
                if handle is not None:
                    L.uz_graph_destroy(handle)
        self._graphs.clear()

    def _run_backward(self, plan, gout):
        plan.loss_scale_t.copy_(gout.reshape(1).to(torch.float32))
        # torch semantics: a second backward() without zero_grad() ACCUMULATES into .grad.  The tape overwrites its
        # regions of the flat gradient buffer, so in that (rare) case the previous buffer is set aside and added back.
        gflat = self._ptab.gflat
        lo, hi = gflat.data_ptr(), gflat.data_ptr() + 4 * gflat.numel()
        prev = None
        if any(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in self._pmap.values()):
            prev = gflat.clone()
            gflat.zero_()
        self._run(plan, "bwd")
        if prev is not None:
            gflat.add_(prev)
        self._post_backward(plan)
        if getattr(self, "_dp", None) is not None:
            self._dp.sync(plan, serial=prev is not None)
        for key in plan.param_grads:
            p = self._pmap[key]
            g = self._ptab.gview(key)
            pg = p.grad
            if pg is None or pg is g or pg.data_ptr() == g.data_ptr():
                p.grad = g
            else:
                p.grad.add_(g)

    def _post_backward(self, plan):
        if plan.__dict__.get("bn_prefixes_nbt_bwd"):           # reversible blocks re-ran their BatchNorms while recomputing
            self._bump_nbt(plan, "bn_prefixes_nbt_bwd")

    def zero_grad(self, set_to_none=True):
        for p in self._pmap.values():
            p.grad = None

    def _loss_tensor(self, plan):
        return _TapeLoss.apply(self._anchor, self, plan)

    def _bump_nbt(self, plan, which="bn_prefixes_nbt"):
        """num_batches_tracked += 1 for every BatchNorm the pass ran (uz_step_counters: one tiny launch of our own, no ATen
        index_add_ in the step)."""
        idx = plan.__dict__.get("_nbt_idx_" + which)
        if idx is None:
            ks = [self._ptab.nbt_keys.index(k) for k in plan.__dict__.get(which, [])]
            idx = torch.tensor(ks, dtype=torch.int64, device=self.device)
            plan.__dict__["_nbt_idx_" + which] = idx
        if idx.numel():
            _ffi.check(_ffi.lib().uz_step_counters(self._ptab.nbt.data_ptr(), idx.data_ptr(), idx.numel(), None, 0, C.c_void_p(self._stream())),
                       "step_counters")

    def _rng_state(self):
        """{seed, offset} of the device-side latent-noise stream (uz_randn_fill): seeded from torch's default generator when
        first used, so torch.manual_seed(...) before the first draw makes runs repeatable.  Under data parallelism the rank is
        mixed into the Philox key: ranks that share a torch seed still draw different noise for their different shards."""
        st = self.__dict__.get("_rng_state_t")
        if st is None:
            rank = int(os.environ.get("RANK", "0")) if getattr(self, "_dp", None) is not None or int(os.environ.get("WORLD_SIZE", "1")) > 1 else 0
            seed = (torch.initial_seed() ^ (rank << 32) ^ (rank * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF
            st = torch.tensor([seed, 0], dtype=torch.int64, device=self.device)
            object.__setattr__(self, "_rng_state_t", st)
        return st

    def rng_state(self):
        """(seed, offset) of the latent-noise stream as host integers - what a resumed run hands to set_rng_state() to continue the
        sequence instead of replaying it from offset 0 (the reference's checkpoints hold the state_dict only, train_model.py:558-564;
        state_dict keys are part of the drop-in surface, so the stream is not stored there)."""
        s = self._rng_state().cpu()
        return int(s[0]), int(s[1])

    def set_rng_state(self, seed, offset=0):
        self._rng_state().copy_(torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, int(offset)], dtype=torch.int64))

    def _fill_normal(self, t):
        """t.normal_() without ATen: Philox / Box-Muller on the device, the stream offset advanced behind the fill."""
        assert t.dtype == torch.float32
        if not t.is_contiguous():                       # (padded volume views): draw into a dense scratch, then move
            return t.copy_(self._fill_normal(torch.empty(t.shape, dtype=torch.float32, device=t.device)))
        L, st, s = _ffi.lib(), self._rng_state(), C.c_void_p(self._stream())
        _ffi.check(L.uz_randn_fill(t.data_ptr(), t.numel(), st.data_ptr(), s), "randn_fill")
        _ffi.check(L.uz_step_counters(None, None, 0, st.data_ptr(), (t.numel() + 3) // 4, s), "step_counters")
        return t


def conv_unit(plan, x, prefix, out=None, relu=True, recompute=False, **kw):
    """Reference Conv2D unit addressed by its module prefix (`<prefix>.convolution.{0,1}`).  recompute=True: the unit is
    re-run inside the backward tape (reversible blocks) - its BatchNorm counts another tracked batch there."""
    a = plan.conv_bn_relu(x, prefix + ".convolution.0", prefix + ".convolution.1", out=out, relu=relu, **kw)
    if recompute:
        which = "bn_prefixes_nbt_bwd"
    elif plan.target is plan.fwd_ops:
        which = "bn_prefixes_nbt"
    elif plan.in_loss_phase():
        which = "bn_prefixes_nbt_loss"
    else:
        which = "bn_prefixes_nbt_extra"
    plan.__dict__.setdefault(which, []).append(prefix + ".convolution.1.num_batches_tracked")
    return a
