"""Data-parallel plumbing: one process per GPU.

The hot path shards by samples: every rank holds a full replica and runs forward / backward on its own images; the only
data-path exchange is the average of the flat fp32 gradient buffer (98 MB for PHiSeg 7/5), identical to the full-batch
mean-loss gradient for equal shard sizes (BatchNorm statistics stay per replica, as in DDP).

Two planes:
  * control plane - a gloo process group on the HOST (init_from_env): the RCCL unique-id hand-off, barriers, the maximum over
    ranks of the timed region, the scheduler's mean loss (mean_scalar) and collective decisions (max_int);
  * data plane - ONE RCCL communicator per process, opened through the C ABI (uz_comm_init, csrc/comm.hip) by GradSync:
    parameters are broadcast and gradients averaged on it, bucket by bucket (one bucket per sub-network), each bucket as soon
    as the backward tape has finished it, on a communication stream beside the remaining backward kernels.
    backend="torch" swaps the data plane for torch.distributed collectives (the gloo test double of the CPU tier).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, local_rank, world_size); a single process needs no group."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    elif world > 1 and torch.cuda.is_available() and dist.get_backend() == "nccl":
        torch.cuda.set_device(local_rank)
    return rank, local_rank, world


def mean_scalar(t, group=None):
    """Mean of a 0-d tensor over the ranks (identity for a single process)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    v = t.detach().clone().reshape(1).to(torch.float32)
    if dist.get_backend(group) == "gloo":
        v = v.cpu()                              # host control plane: the scheduler compares on the host anyway (one sync per step)
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    return (v / dist.get_world_size(group)).reshape(()).to(t.device)


def max_int(v, group=None):
    """Maximum of a host integer over the ranks (identity for a single process): collective decisions - e.g. "did ANY rank see
    a violated magnitude bound" - must come out the same everywhere, or the ranks' collectives stop pairing up."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return int(v)
    t = torch.tensor([int(v)], dtype=torch.int64)
    if dist.get_backend(group) != "gloo":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def or_flags(word, group=None, bits=3):
    """Bitwise OR of a small host flag word over the ranks in ONE collective (identity for a single process): the bits travel as a
    vector of 0 / 1 and are combined with MAX (every backend has it; BOR is not available on all of them)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return int(word)
    t = torch.tensor([(int(word) >> b) & 1 for b in range(bits)], dtype=torch.int64)
    if dist.get_backend(group) != "gloo":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return sum(int(v) << b for b, v in enumerate(t.tolist()))


def max_over_ranks(values, group=None):
    """Element-wise MAX of a list of host floats over the ranks in one collective (identity for a single process): every rank gets the
    same list back, so decisions taken from it (Engine.tune_schedule: per-op durations, tape times, which schedule stays) are the same
    on every rank - the bucket exchange order follows the scheduled tape and must not differ between ranks."""
    vals = [float(v) for v in values]
    if not dist.is_initialized() or dist.get_world_size(group) == 1 or not vals:
        return vals
    t = torch.tensor(vals, dtype=torch.float64)
    if dist.get_backend(group) != "gloo":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.tolist()


def shard_bounds(n, rank, world):
    """Contiguous [lo, hi) slice of n samples owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_mean_(flat, group=None):
    """In-place average of a flat gradient buffer over the ranks of `group`."""
    if not dist.is_initialized():
        return flat
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / world)
    return flat


def broadcast_(flat, src=0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


# ----------------------------------------------------------------------------------------------
# Bucketed, overlapped gradient exchange
# ----------------------------------------------------------------------------------------------
def param_buckets(ptab, target_floats=None):
    """Contiguous [lo, hi) float ranges of the flat gradient buffer, cut at tensor boundaries into BYTE-WEIGHTED slices of about
    `target_floats` (default 3 Mi floats = 12 MB; UZ_DP_BUCKET_MB overrides).  Rounds 2 - 5 used one bucket per sub-network
    (PHiSeg: 37.7 / 22.6 / 37.7 MB): under the lane replay the backward tape finishes ALL of them in its last 3 % (the weight
    gradients of the deep levels - 80 % of the bytes - are the lanes' fillers), so the whole exchange trailed the tape and the LAST
    bucket alone was 38 % of the bytes.  With slices the last-final bucket is one slice (12 %), the others leave as the tape
    finalises them, in the order of the scheduled tape (GradSync.order).  A ring all-reduce below a few MB is latency, not
    bandwidth: slices are not cut finer than that, and a short tail is merged into its neighbour."""
    if target_floats is None:
        target_floats = int(float(os.environ.get("UZ_DP_BUCKET_MB", "12")) * (1 << 20) / 4)
    out, lo, acc = [], 0, 0
    end = 0
    for key, off in ptab.poff.items():
        n = 1
        for s_ in ptab.shape[key]:
            n *= int(s_)
        acc += n
        end = off + n
        if acc >= target_floats:
            out.append([lo, end])
            lo, acc = end, 0
    if lo < end:
        if out and end - lo < target_floats // 2:
            out[-1][1] = end
        else:
            out.append([lo, end])
    return [(a_, b_) for a_, b_ in out]


class GradSync:
    """Averages the flat gradient buffer over the ranks, one bucket at a time, overlapped with the backward tape.

    backend "rccl": RCCL through this package's own C ABI (uz_comm_* over librccl.so) on a private high-priority
    communication stream; torch.distributed is used ONCE, to hand rank 0's 128-byte unique id to the other ranks.
    The backward tape records one event per bucket (UZ_OP_EVENT_RECORD, also inside the hipGraph); per bucket the
    communication stream waits for that event and runs ncclAllReduce(avg) while the compute stream continues; the
    compute stream then waits for ONE event behind the last all-reduce, so Adam starts as soon as the gradients are
    averaged.  No host synchronisation anywhere.
    backend "torch": torch.distributed collectives (gloo on a one-GPU test box, where RCCL refuses two ranks per
    device): same buckets, same order, blocking."""

    def __init__(self, model, group=None, backend=None, overlap=True):
        import ctypes as C
        from . import _ffi
        self.model, self.group, self.overlap = model, group, bool(overlap)
        self.C, self._ffi, self.L = C, _ffi, _ffi.lib()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if backend is None:
            backend = os.environ.get("UZ_DP_BACKEND") or \
                ("rccl" if (dist.is_initialized() and dist.get_backend(group) == "nccl") else "torch")
        if backend not in ("rccl", "torch"):
            raise ValueError(f"GradSync backend {backend!r}: expected 'rccl' or 'torch'")
        self.backend = backend
        self.comm = self.stream = self.done = None
        self.t0 = self.t1 = None
        self.buckets = param_buckets(model._ptab)
        if backend == "rccl":
            self._init_rccl()

    # ------------------------------------------------------------------ RCCL through the C ABI
    def _init_rccl(self):
        C, L, check = self.C, self.L, self._ffi.check
        lib = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")     # the copy this process already maps
        check(L.uz_comm_load(lib.encode() if os.path.exists(lib) else None), "comm_load")
        uid = (C.c_char * 128)()
        if self.rank == 0:
            check(L.uz_comm_unique_id(uid), "comm_unique_id")
        if self.world > 1:
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        check(L.uz_comm_init(self.rank, self.world, uid, C.byref(h)), "comm_init")
        self.comm = h.value
        s = C.c_void_p()
        check(L.uz_stream_create(C.byref(s), int(os.environ.get("UZ_DP_STREAM_PRIORITY", "1"))), "stream_create")
        self.stream = s.value
        ev = []
        for timing in (0, 1, 1) + (1,) * len(self.buckets):
            e = C.c_void_p()
            check(L.uz_event_create(C.byref(e), timing), "event_create")
            ev.append(e.value)
        self.done, self.t0, self.t1 = ev[:3]
        self.tb = ev[3:]                 # one timing event per bucket: recorded on the communication stream behind that bucket's all-reduce
        self._tb_order = []

    def nranks(self):
        """Ranks of the data-parallel group as the communication library itself reports them (ncclCommCount behind
        uz_comm_size for RCCL; the process group's size for the torch test double)."""
        if self.backend == "rccl":
            return int(self.L.uz_comm_size(self.comm))
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def broadcast_params(self):
        """Replicas start from rank 0's parameters and BatchNorm buffers."""
        pt = self.model._ptab
        if self.world == 1:
            return
        if self.backend == "rccl":
            st = self.model._stream()
            self._ffi.check(self.L.uz_broadcast_f32(self.comm, pt.pflat.data_ptr(), pt.n_params, 0, st), "broadcast")
            self._ffi.check(self.L.uz_broadcast_f32(self.comm, pt.bflat.data_ptr(), pt.n_buffers, 0, st), "broadcast")
        else:
            broadcast_(pt.pflat, 0, self.group)
            broadcast_(pt.bflat, 0, self.group)

    def order(self, plan):
        """Bucket indices in the order the (scheduled) backward tape finishes them."""
        pos = {}
        for k, o in enumerate(plan.bwd_ops):
            if o["code"] == "UZ_OP_EVENT_RECORD":
                pos[o["p"][0][1]] = k
        return sorted(pos, key=pos.get)

    def sync(self, plan, serial=False):
        """Called right after the backward tape has been launched on the compute stream.  serial=True: the caller
        touched the gradient buffer after the tape (gradient accumulation), so the bucket events are stale - exchange
        everything behind the compute stream's current position instead."""
        gflat = self.model._ptab.gflat
        base = gflat.data_ptr()
        if self.backend != "rccl":
            if self.world > 1:
                for b in (self.order(plan) if plan.events else range(len(self.buckets))):
                    lo, hi = self.buckets[b]
                    dist.all_reduce(gflat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
                gflat.mul_(1.0 / self.world)
            return
        L, check, cs = self.L, self._ffi.check, self.model._stream()
        if not cs and self.overlap and not serial:
            # The legacy default (NULL) stream serialises against other streams' dependencies: measured on MI355X / ROCm 7.2,
            # a second stream merely WAITING on events recorded between the backward graphs costs +6 ms per step there and the
            # overlapped all-reduce +15 ms, against +0.2 ms on any created stream.  Run data-parallel training under
            # torch.cuda.stream(...) / torch.cuda.set_stream(...) (bench.py and train_model.py do); on the NULL stream fall
            # back to one exchange behind the whole tape.
            if not getattr(self, "_warned_null", False):
                import warnings
                warnings.warn("data-parallel overlap is disabled on the legacy default stream; make a created stream current "
                              "(torch.cuda.set_stream(torch.cuda.Stream())) to overlap the gradient all-reduce with backward")
                self._warned_null = True
            serial = True
        check(L.uz_event_record(self.t0, cs), "event_record")              # compute stream: backward tape done
        self._tb_order = []
        if self.overlap and plan.events and not serial:
            for b in self.order(plan):
                lo, hi = self.buckets[b]
                check(L.uz_stream_wait_event(self.stream, plan.events[b]), "stream_wait_event")
                check(L.uz_allreduce_mean_f32(self.comm, base + 4 * lo, hi - lo, self.stream), "allreduce")
                check(L.uz_event_record(self.tb[b], self.stream), "event_record")
                self._tb_order.append(b)
        else:                                                                # one blocking-order all-reduce behind the whole tape
            check(L.uz_stream_wait_event(self.stream, self.t0), "stream_wait_event")
            check(L.uz_allreduce_mean_f32(self.comm, base, gflat.numel(), self.stream), "allreduce")
        check(L.uz_event_record(self.t1, self.stream), "event_record")
        check(L.uz_stream_wait_event(cs, self.t1), "stream_wait_event")    # Adam (compute stream) starts behind the last all-reduce

    def exposed_ms(self):
        """Time between the end of the backward tape and the end of the last all-reduce of the most recent step
        (= what the collective adds to the step; <= 0 means fully hidden).  Synchronises."""
        if self.backend != "rccl" or self.t0 is None:
            return None
        torch.cuda.synchronize()
        ms = self.C.c_float()
        self._ffi.check(self.L.uz_event_elapsed_ms(self.t0, self.t1, self.C.byref(ms)), "event_elapsed")
        return float(ms.value)

    def exposed_ms_per_bucket(self):
        """For every bucket of the most recent overlapped step, in exchange order: (first float, floats, milliseconds between the end
        of the backward tape and the end of that bucket's all-reduce) - negative = the exchange finished that long BEFORE the tape
        did (fully hidden); the last bucket's positive value is what the collective adds to the step.  Synchronises."""
        if self.backend != "rccl" or not self._tb_order:
            return None
        torch.cuda.synchronize()
        out = []
        for b in self._tb_order:
            ms = self.C.c_float()
            # hipEventElapsedTime(t0, tb) is negative when tb was recorded first
            self._ffi.check(self.L.uz_event_elapsed_ms(self.t0, self.tb[b], self.C.byref(ms)), "event_elapsed")
            lo, hi = self.buckets[b]
            out.append(dict(first_float=int(lo), floats=int(hi - lo), ms_after_backward=round(float(ms.value), 4)))
        return out

    def close(self):
        """Destroys the communicator AND the communication stream: a process that builds several GradSyncs one after the other (a test
        harness; tools/nccl_world1_check.py) otherwise keeps every earlier communication stream alive - streams share the process's
        GPU_MAX_HW_QUEUES hardware queues with the replay's lanes, and the third model of such a process ran its step in 21.7 ms instead
        of 16.1 (round 6: the "--no-overlap pathology" of the last review was this, not the exchange)."""
        if self.comm:
            self.L.uz_comm_destroy(self.comm)
            self.comm = None
        if self.stream:
            self.L.uz_stream_destroy(self.stream)
            self.stream = None
