"""Static execution plan ("tape") builder.

A model instance flattens itself, for a given (batch, H, W, mode), into three lists of C-ABI
calls - forward, loss, backward - over pre-allocated device buffers.  Shapes are static, so the
lists are built once, turned into ``uz_op`` arrays and replayed with one FFI call per pass
(``uz_run_tape``) or as a captured hipGraph.  Nothing here computes on the host: the plan only
decides *which* kernel runs on *which* buffer.

Design points:
  * concat-free: a producer writes straight into a channel slice of the consumer's buffer
    (``View`` = buffer + channel offset), replacing torch.cat (phiseg.py:71,183,315; unet.py:72);
  * gradient fan-in is resolved at plan time: the first backward writer of a gradient region
    overwrites, later writers accumulate (no zero-fill passes);
  * branches whose output never reaches the loss are skipped in backward, which reproduces the
    reference's ``grad is None`` parameters.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _ffi

BN_EPS = 1e-3        # reference torchlayers.py:20
BN_MOMENTUM = 0.01   # reference torchlayers.py:20
_ALIGN = 64          # floats
_AMAX_FLOATS = 256   # floats per magnitude-bound slot: 16 sub-slots 64 bytes apart (csrc/split_f16.h)


def _numel(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


# ----------------------------------------------------------------------------------------------
class ParamTable:
    """Flat fp32 storage for all parameters / BN buffers of a model, addressed by the
    reference's state_dict keys.  spec: ordered [(key, shape, kind)], kind in
    conv_w conv_b bn_w bn_b bn_rm bn_rv bn_nbt (same vocabulary as the golden fixtures)."""

    def __init__(self, spec, device):
        self.spec = [(k, tuple(s), kd) for k, s, kd in spec]
        self.device = device
        self.poff, self.boff, self.shape, self.kind = OrderedDict(), OrderedDict(), {}, {}
        po = bo = 0
        self.nbt_keys = []
        for k, s, kd in self.spec:
            self.shape[k], self.kind[k] = s, kd
            if kd in ("conv_w", "conv_b", "bn_w", "bn_b"):
                self.poff[k] = po
                po += _numel(s)
            elif kd in ("bn_rm", "bn_rv"):
                self.boff[k] = bo
                bo += _numel(s)
            elif kd == "bn_nbt":
                self.nbt_keys.append(k)
            else:
                raise ValueError(kd)
        self.n_params, self.n_buffers = po, bo
        self.pflat = torch.zeros(max(po, 1), dtype=torch.float32, device=device)
        self.gflat = torch.zeros(max(po, 1), dtype=torch.float32, device=device)
        self.bflat = torch.zeros(max(bo, 1), dtype=torch.float32, device=device)
        self.nbt = torch.zeros(max(len(self.nbt_keys), 1), dtype=torch.int64, device=device)

    def pview(self, k):
        o = self.poff[k]
        return self.pflat[o:o + _numel(self.shape[k])].view(self.shape[k])

    def gview(self, k):
        """View of parameter k's gradient in the flat buffer.  The view OBJECTS are cached (the buffer never moves): a backward pass
        re-attaches ~400 of them and the optimiser compares them by identity - slicing them anew cost 2 ms of host time per step."""
        cache = self.__dict__.setdefault("_gviews", {})
        v = cache.get(k)
        if v is None:
            o = self.poff[k]
            v = cache[k] = self.gflat[o:o + _numel(self.shape[k])].view(self.shape[k])
        return v

    def bview(self, k):
        o = self.boff[k]
        return self.bflat[o:o + _numel(self.shape[k])].view(self.shape[k])

    def nbtview(self, k):
        return self.nbt[self.nbt_keys.index(k)]


class Buf:
    __slots__ = ("name", "N", "C", "H", "W", "off", "gbuf", "requires_grad", "amax", "amax_cov", "alias", "relu", "packed", "b16", "vol")

    def __init__(self, name, N, C, H, W, requires_grad=True):
        self.name, self.N, self.C, self.H, self.W = name, int(N), int(C), int(H), int(W)
        self.off, self.gbuf, self.requires_grad = None, None, requires_grad
        self.amax, self.amax_cov = None, []      # magnitude-bound slot and the channel ranges whose producers maintain it
        self.alias = None                        # pooled scratch: several differently shaped Bufs share one arena region
        self.relu = None                         # set by Plan.conv_relu: {(c0, C): state} of Conv -> ReLU units writing into this buffer
        self.packed = False                      # kept in split storage (Plan._round4_passes): Plan.tensor() of it is NOT fp32 values
        self.b16 = False                         # kept in bf16 storage (Plan._b16_pass): 2-byte elements, Plan.tensor() of it is a bfloat16 tensor
        self.vol = False                         # a volume (Plan.vol): [D + 2][C][H][W]

    @property
    def numel(self):
        return self.N * self.C * self.H * self.W

    @property
    def words(self):
        """32-bit words of arena the buffer occupies."""
        return (self.numel + 1) // 2 if self.b16 else self.numel

    @property
    def esz(self):
        return 2 if self.b16 else 4


class View:
    """Channel slice [c0, c0+C) of a buffer, optionally restricted to the batch entries [b0, b0+nb) (volumes: the D real
    slices between the two zero slices, see Plan.vol)."""
    __slots__ = ("buf", "c0", "C", "b0", "nb")

    def __init__(self, buf, c0=0, C=None, b0=0, nb=None):
        self.buf, self.c0, self.C = buf, c0, buf.C - c0 if C is None else C
        self.b0, self.nb = b0, nb
        assert 0 <= self.c0 and self.c0 + self.C <= buf.C

    N = property(lambda s: s.buf.N if s.nb is None else s.nb)
    H = property(lambda s: s.buf.H)
    W = property(lambda s: s.buf.W)
    Ctot = property(lambda s: s.buf.C)

    def slice(self, c0, C):
        return View(self.buf, self.c0 + c0, C, self.b0, self.nb)

    @property
    def contiguous(self):
        return self.c0 == 0 and self.C == self.buf.C

    @property
    def numel(self):
        return self.N * self.C * self.H * self.W


class Latent:
    """One (mu, sigma, z) head: SampleZBlock tail (phiseg.py:99-106) or AxisAlignedConvGaussian
    (probabilistic_unet.py:124-129)."""
    def __init__(self, mu, pre, sigma, z, eps, act):
        self.mu, self.pre, self.sigma, self.z, self.eps, self.act = mu, pre, sigma, z, eps, act
        self.kl_dmu = self.kl_dsigma = None


# ----------------------------------------------------------------------------------------------
class Plan:
    def __init__(self, N, ptab, bn_training, device):
        self.N, self.ptab, self.bn_training, self.device = int(N), ptab, bool(bn_training), device
        self.codes = _ffi.op_codes()
        self.L = _ffi.lib()
        self.bufs = []
        self.fwd_ops, self.loss_ops, self.bwd_ops = [], [], []
        self.extra_ops = {}                 # additional forward-only tapes (e.g. "decode")
        self.target = self.fwd_ops
        self._bwd = []                      # closures, run in reverse
        self._bwd_tail = []                 # closures run after all others (regulariser gradients)
        self._bwd_head = []                 # closures run before all others (re-materialised backward inputs)
        self._record_bwd = True
        self._ginit = {}                    # Buf -> list of (c0, c1) gradient regions already written
        self.scratch = dict(bn=0, wgrad=0, gy=0, ce=0)   # bytes (bn, wgrad, ce) / floats (gy)
        self.ptr_tables = []                # list of lists of pointer refs -> device int64 table
        self.param_grads = []               # parameter keys that receive a gradient, in write order
        self.named = {}                     # user-visible tensors: name -> View
        self.finalized = False
        # Magnitude bounds for the split-fp16 convolutions: slot 0 bounds |parameters| (measured at the head of the forward
        # tape), one slot per activation buffer (atomic-max maintained by the kernels that write it) and one per Conv unit for
        # the gradient w.r.t. its output.  Forward-side slots are zeroed at the head of the forward tape, backward-side slots
        # at the head of the backward tape.
        self.n_amax, self.n_amax_fwd = 1, None
        self._amax_bwd = []                 # slot indices used by backward producers
        self._packs = {"fwd": {}, "bwd": {}}   # pre-packed split-fp16 weight images: wkey -> geometry (see _packed)
        self._packbuf = {}
        self.grad_buckets = []              # [(lo, hi)] float ranges of the flat gradient buffer that get a "final" event (data parallel)
        self.events = []                    # hipEvent_t handles (uz_event_create), one per gradient bucket
        self._gid = 0                       # scheduling group of the ops being emitted (see _schedule)
        self._gyz = {}                       # zero-bordered dy scratch classes of volume units: (slices, floats per slice) -> floats
        self.n_lanes = max(1, min(int(os.environ.get("UZ_LANES", "2")), 8))
        self.decouple_wgrad_px = 0           # NativeModel.decouple_wgrad_px: planes (N*H*W) up to which weight gradients get a group of their own
        self.decouple_wgrad_prefixes = ()    # NativeModel.decouple_wgrad_prefixes
        self._nclaims = {}                   # backward writers seen per buffer (conv_relu's folded ReLU backward checks it was the last one)
        self._dbias_jobs = []                # folded ReLU backward: bias gradients still to be summed from their partials (one launch at the tape's end)
        self.chain_px = 0                    # NativeModel.chain_px: planes (N*H*W) up to which a tape's ops run as phases of ONE persistent launch (_chain_pass)
        self.chain_wgs = 256                 # workgroups of that launch (forward tape: nothing else runs beside it)
        self.chain_wgs_bwd = 80              # ... of a backward chain (it runs beside the other lanes' device-filling kernels; at most three chains at once)
        self._chains = []                    # per chain op: dict(sub=[...], phases=[...]) -> device tables built on first resolve
        self._chain_imgs = {}                # tape -> {wkey: dict(off, bytes, mc, kc, cin, dgrad)} packed weight images of the chain convolutions
        self._chain_imgbuf = {}

    def _newgroup(self):
        self._gid += 1

    # ------------------------------------------------------------------ buffers
    def buf(self, name, C, H, W, N=None, requires_grad=True):
        b = Buf(name, self.N if N is None else N, C, H, W, requires_grad)
        self.bufs.append(b)
        return View(b)

    def vec(self, name, n, requires_grad=False):
        """Small 1-D float region (loss terms, saved BN statistics, ...)."""
        return self.buf(name, n, 1, 1, N=1, requires_grad=requires_grad)

    def gview(self, v):
        if v.buf.gbuf is None:
            g = Buf("grad:" + v.buf.name, v.buf.N, v.buf.C, v.buf.H, v.buf.W, False)
            g.vol = v.buf.vol
            self.bufs.append(g)
            v.buf.gbuf = g
        return View(v.buf.gbuf, v.c0, v.C, v.b0, v.nb)

    def vol(self, name, C, D, H, W, requires_grad=True):
        """One volume (models/phiseg3D.py): [D + 2][C][H][W] with a zero slice on either side of the D real ones (csrc/vol.hip);
        the returned view addresses the real slices as a batch of D images.  The arena is zero-initialised and nothing ever
        writes the two border slices, so they ARE the depth padding of the 3x3x3 convolutions."""
        b = Buf(name, D + 2, C, H, W, requires_grad)
        b.vol = True
        self.bufs.append(b)
        return View(b, 0, C, 1, D)

    # ------------------------------------------------------------------ magnitude bounds
    def _new_amax(self, bwd=False):
        k = self.n_amax
        self.n_amax += 1
        if bwd:
            self._amax_bwd.append(k)
        return k

    def amax_out(self, v):
        """Slot reference a producer of view `v` accumulates max|v| into (creates the buffer's slot on first use)."""
        if isinstance(v, _ScratchView):
            return ("amax", v.amax, "acc") if v.amax is not None else None
        b = v.buf
        if b.amax is None:
            b.amax = self._new_amax()
        b.amax_cov.append((v.c0, v.c0 + v.C))
        return ("amax", b.amax, "acc")       # third element: this op ACCUMULATES into the slot (see _access)

    def amax_in(self, v):
        """Slot reference bounding |v| for a consumer, or None when some producer of its channels keeps no bound
        (the kernel then measures the tensor itself)."""
        if isinstance(v, _ScratchView):
            return ("amax", v.amax) if v.amax is not None else None
        b = v.buf
        if b.amax is None:
            return None
        ok = all(any(a <= c < e for a, e in b.amax_cov) for c in range(v.c0, v.c0 + v.C))
        return ("amax", b.amax) if ok else None

    def _claim(self, v):
        """Returns the accumulate flag for a backward write to grad(v) and records the region."""
        regs = self._ginit.setdefault(v.buf, [])
        self._nclaims[v.buf] = self._nclaims.get(v.buf, 0) + 1
        lo, hi = v.c0, v.c0 + v.C
        covered = [c for c in range(lo, hi) if any(a <= c < b for a, b in regs)]
        if len(covered) == hi - lo:
            return 1
        if covered:
            raise RuntimeError(f"partial gradient overlap on {v.buf.name}[{lo}:{hi}] vs {regs}")
        regs.append((lo, hi))
        return 0

    def _has_grad(self, v):
        regs = self._ginit.get(v.buf, [])
        return all(any(a <= c < b for a, b in regs) for c in range(v.c0, v.c0 + v.C))

    # ------------------------------------------------------------------ op emission
    def _emit(self, lst, code, p=(), i=(), f=(), n=0, detached=False):
        """detached: a scheduling group of its own, whatever is emitted around it (parameter-only preparation ops: they are ready at
        the head of the tape and must not sit in a layer's chain)."""
        gid = self._gid
        if detached:
            self._detached = getattr(self, "_detached", 0) + 1
            gid = ("detached", self._detached)
        op = dict(code=code, p=list(p), i=[int(x) for x in i], f=[float(x) for x in f], n=int(n), gid=gid)
        lst.append(op)
        return op

    def P(self, key, extra=0):
        return ("param", key, extra)

    def G(self, key, extra=0):
        if key not in self.param_grads:
            self.param_grads.append(key)
        return ("pgrad", key, extra)

    def B(self, key):
        return ("buffer", key, 0)

    def loss_phase(self):
        self.target = self.loss_ops

    def extra_phase(self, name):
        """Switch to an extra forward-only tape (no backward is recorded for its ops)."""
        self.target = self.extra_ops.setdefault(name, [])
        self._record_bwd = False

    def _push_bwd(self, fn):
        if self._record_bwd:
            self._bwd.append(fn)

    def in_loss_phase(self):
        return self.target is self.loss_ops

    def _tables_ok(self):
        """Deferred table-driven reductions (weight-gradient slabs, bias-gradient rows).  Data parallel: a bucket's gradients must be
        final when its all-reduce starts, so there the tables are cut per BUCKET (finalize(): one launch per bucket, and the bucket's
        "final" marker depends on it through the gradient ranges it writes); UZ_DP_TABLES=0 restores one reduction launch per layer."""
        return not self.grad_buckets or os.environ.get("UZ_DP_TABLES", "1") == "1"

    def _by_bucket(self, jobs, key_of):
        """jobs split by the gradient bucket their parameter lies in (one group when there are no buckets), emission order kept."""
        if not self.grad_buckets:
            return [jobs]
        groups = [[j for j in jobs if lo <= self.ptab.poff[key_of(j)] < hi] for lo, hi in self.grad_buckets]
        assert sum(len(g) for g in groups) == len(jobs), "a deferred reduction's parameter lies outside every gradient bucket"
        return [g for g in groups if g]

    def l2_reg(self, keys, term, coeff):
        """coeff * sum_k ||p_k||_2 over the listed parameter tensors (utils.l2_regularisation, utils.py:93-101)."""
        n = len(keys)
        tab = self.ptr_table([("raw", v) for k in keys for v in (self.ptab.poff[k], _numel(self.ptab.shape[k]))])
        norms = self.vec("l2_norms", n)
        self._newgroup()
        self._emit(self.target, "UZ_OP_L2_NORMS", p=[("pflat",), tab, norms], i=[n])
        self._emit(self.target, "UZ_OP_SUM_TERMS", p=[norms, term], i=[n])
        self._emit(self.target, "UZ_OP_SCALE", p=[term], f=[coeff], n=1)

        def tail():
            sc = self.vec("l2_scale", 1)
            self._emit(self.bwd_ops, "UZ_OP_COPY", p=[sc, self.loss_scale], n=4)
            self._emit(self.bwd_ops, "UZ_OP_SCALE", p=[sc], f=[coeff], n=1)
            # the write set is the listed tensors' gradient ranges only: with the whole flat buffer declared, every
            # gradient bucket's "final" marker waited for this op and the data-parallel overlap collapsed to one exchange
            self._emit(self.bwd_ops, "UZ_OP_L2_NORMS_BWD", p=[("pflat",), tab, norms, sc, ("gflat_keys", tuple(keys))], i=[n])
        self._bwd_tail.append(tail)

    # ------------------------------------------------------------------ convolution family
    def _conv_fwd(self, x, wkey, bkey, y, ks, relu, wrow0=0, bn_partials=None, slabs_only=False):
        cout = y.C
        cin = x.C
        if len(self.ptab.shape[wkey]) == 5 and x.nb is None:
            raise RuntimeError(f"{wkey} is a Conv3d weight but its input {x.buf.name} is not a volume")
        if x.nb is not None and ks == 3:
            # Conv3d(3x3x3, pad 1) on a volume = the 2-D kernel over the D slices with the depth window as 3 Cin input channels
            # (csrc/vol.hip); the weights are permuted to [co][kd][ci][3][3] first
            assert wrow0 == 0
            x = self._window_input(x, wkey)
            ws = self.L.uz_conv_workspace(3 * cin, cout, x.N, x.H, x.W, 3)
            self.scratch["wgrad"] = max(self.scratch["wgrad"], ws)
            self._newgroup()
            packed = self._packed(wkey, cin, cout, x, False, vol=True)
            wp = self.P(wkey)                                  # (unused by the kernel when the image is pre-packed)
            if packed is None:
                wp = self.vec(wkey + ":w3d_fwd", cout * cin * 27)
                self._emit(self.target, "UZ_OP_W3D_PERMUTE", p=[self.P(wkey), wp], i=[cout, cin, 0], detached=True)
            self._emit(self.target, "UZ_OP_CONV_FWD",
                       p=[("win", x), wp, self.P(bkey) if bkey else None, y, ("scratch", "wgrad"), self.amax_in(x), ("amax", 0),
                          self.amax_out(y) if relu else None, packed, bn_partials],
                       i=[3 * cin, x.Ctot, cout, y.Ctot, x.N, x.H, x.W, 3, relu], n=ws)
            return
        wextra = wrow0 * cin * ks * ks
        ws = self.L.uz_conv_workspace(cin, cout, x.N, x.H, x.W, ks)
        self.scratch["wgrad"] = max(self.scratch["wgrad"], ws)
        self._newgroup()
        packed = self._packed(wkey, cin, cout, x, False) if (ks == 3 and wrow0 == 0) else None
        self._emit(self.target, "UZ_OP_CONV_FWD",
                   p=[x, self.P(wkey, wextra), self.P(bkey, wrow0) if bkey else None, y, ("scratch", "wgrad"),
                      self.amax_in(x), ("amax", 0), self.amax_out(y) if (relu and ks == 3) else None, packed, bn_partials],
                   i=[cin, x.Ctot, cout, y.Ctot, x.N, x.H, x.W, ks, relu, int(slabs_only)], n=ws)

    def _conv_bwd(self, x, wkey, gy, ks, db_key=None, wrow0=0, rec=None):
        """Weight gradient (+ optional bias gradient) and, when the input carries a gradient,
        the data gradient.  gy: gradient w.r.t. the conv output (a View).  rec (dict): receives the emitted 2-D ops
        ("wgrad", "dgrad") for the round-4 passes of finalize()."""
        rec = rec if rec is not None else {}
        cin, cout = x.C, gy.C
        if x.nb is not None and ks == 3:                      # volume: see _conv_fwd
            assert isinstance(gy, _ScratchView) and gy.off and db_key is None and wrow0 == 0
            x, x_orig = self.__dict__.get("_win_of", {}).get(wkey, x), x          # the weight gradient reads the depth window
            ws = self.L.uz_conv_bwd_weight_workspace(3 * cin, cout, x.N, x.H, x.W, 3)
            self.scratch["wgrad"] = max(self.scratch["wgrad"], ws)
            # (a depth-window call - 3 C view channels over a C-channel buffer - writes the Conv3d gradient layout itself:
            # uz_conv_bwd_weight's slab reduction does the [co][kd][ci] -> [co][ci][kd] permutation on the way out)
            # (slab reductions deferred to the table launch like the 2-D layers': the table row carries the window's C)
            nslab = 0
            if self._tables_ok() and os.environ.get("UZ_WGRAD_TABLE", "1") == "1" and os.environ.get("UZ_WGRAD_TABLE_VOL", "1") == "1" \
                    and not self.__dict__.get("_in_rev", False) and self.__dict__.get("_rev_ctx") is None:
                nslab = self.L.uz_conv_bwd_weight_slabs(3 * cin, cout, x.N, x.H, x.W, 3)
            slabbuf = self.vec(wkey + ":wslabs", nslab * 9 * cout * 3 * cin) if nslab else None
            rec["wgrad"] = self._emit(self.bwd_ops, "UZ_OP_CONV_BWD_WEIGHT",
                       p=[("win", x), gy, self.G(wkey), None, ("scratch", "wgrad"), self.amax_in(x), self.amax_in(gy), None, slabbuf],
                       i=[3 * cin, x.Ctot, cout, gy.Ctot, x.N, x.H, x.W, 3, 0, 0, 0, 1 if nslab else 0, nslab], n=ws)      # i[12]: slabs the buffer was sized for
            if nslab:
                self.__dict__.setdefault("_wgrad_jobs", []).append((slabbuf, wkey, nslab, cout, 3 * cin, 9, cin))
            x = x_orig
            if x.buf.requires_grad:
                acc = self._claim(x)
                ws2 = self.L.uz_conv_workspace(cin, 3 * cout, x.N, x.H, x.W, 3)
                self.scratch["wgrad"] = max(self.scratch["wgrad"], ws2)
                packed = self._packed(wkey, cin, cout, x, True, vol=True)
                wp2 = self.P(wkey)
                if packed is None:
                    wp2 = self.vec(wkey + ":w3d_bwd", cout * cin * 27)
                    # the data gradient's weight layout is prepared in the FORWARD tape (same parameters: the optimiser only steps
                    # behind the backward tape), as a group of its own - not in the layer's backward chain
                    self._emit(self.fwd_ops, "UZ_OP_W3D_PERMUTE", p=[self.P(wkey), wp2], i=[cout, cin, 1], detached=True)
                rec["dgrad"] = self._emit(self.bwd_ops, "UZ_OP_CONV_BWD_DATA",
                           p=[("gywin", gy.zkey), wp2, self.gview(x), ("scratch", "wgrad"), self.amax_in(gy), ("amax", 0), packed],
                           i=[3 * cout, gy.Ctot, cin, x.Ctot, x.N, x.H, x.W, 3, acc], n=ws2)
            return
        wextra = wrow0 * cin * ks * ks
        ws = self.L.uz_conv_bwd_weight_workspace(cin, cout, x.N, x.H, x.W, ks)
        self.scratch["wgrad"] = max(self.scratch["wgrad"], ws)

        def emit_wgrad():
            # Slab reductions: outside data-parallel runs (where a bucket's gradients must be final when its all-reduce starts) a
            # 3 x 3 layer leaves its partial-sum slabs in a buffer of its own and ONE table-driven launch at the end of the tape adds
            # the slabs of every layer (UZ_OP_WGRAD_REDUCE_TABLE) - PHiSeg: 106 + 27 small reduction launches less per step
            nslab = 0
            if ks == 3 and db_key is None and wrow0 == 0 and self._tables_ok() and os.environ.get("UZ_WGRAD_TABLE", "1") == "1" \
                    and not self.__dict__.get("_in_rev", False) and self.__dict__.get("_rev_ctx") is None:
                nslab = self.L.uz_conv_bwd_weight_slabs(cin, cout, x.N, x.H, x.W, ks)
            slabbuf = self.vec(wkey + ":wslabs", nslab * ks * ks * cout * cin) if nslab else None
            rec["wgrad"] = self._emit(self.bwd_ops, "UZ_OP_CONV_BWD_WEIGHT",
                                      p=[x, gy, self.G(wkey, wextra), self.G(db_key, wrow0) if db_key else None, ("scratch", "wgrad"),
                                         self.amax_in(x), self.amax_in(gy), None, slabbuf],
                                      i=[cin, x.Ctot, cout, gy.Ctot, x.N, x.H, x.W, ks, 0, 0, 0, 1 if nslab else 0, nslab], n=ws)      # i[12]: slabs the buffer was sized for (checked at launch: ADVICE r5)
            if nslab:
                self.__dict__.setdefault("_wgrad_jobs", []).append((slabbuf, wkey, nslab, cout, cin, ks * ks))
        # dy in a buffer of its own (small planes): the data gradient - the only thing the next layer's backward waits for -
        # goes first and the weight gradient becomes a scheduling group of its own, so that the latency-bound chains of the deep
        # levels (BatchNorm backward -> data gradient -> next BatchNorm backward) no longer carry the weight gradients and their
        # slab reductions; those fill the other dependency lane instead.
        decoupled = isinstance(gy, _ScratchView) and gy.view is not None
        if not decoupled:
            emit_wgrad()
        if x.buf.requires_grad:
            acc = self._claim(x)
            ws2 = self.L.uz_conv_workspace(cin, cout, x.N, x.H, x.W, ks)
            self.scratch["wgrad"] = max(self.scratch["wgrad"], ws2)
            packed = self._packed(wkey, cin, cout, x, True) if (ks == 3 and wrow0 == 0) else None
            # x is the output of a Conv -> ReLU unit (vanilla U-Net blocks) and this data gradient is the last writer of its
            # gradient: fold that unit's ReLU backward into the epilogue (mask, bias-gradient partials, bound) - the unit then
            # uses grad(x) in place as its dy instead of running uz_relu_bwd over it (conv_relu.bwd checks "last writer")
            fold = None
            st = (x.buf.relu or {}).get((x.c0, x.C)) if x.nb is None else None
            if st is not None and ks == 3 and wrow0 == 0 and os.environ.get("UZ_FOLD_RELU_BWD", "1") == "1" \
                    and not self.__dict__.get("_in_rev", False):
                npart = self.L.uz_conv_bwd_relu_partials(cin, cout, x.N, x.H, x.W, ks)
                if npart > 0:
                    fold = st["fold"] = dict(part=self.vec(x.buf.name + f":dbpart{x.c0}", 4 * cin * npart), npart=npart,
                                             amax=self.amax_out(self.gview(x)), claims=self._nclaims[x.buf])
            rec["dgrad"] = self._emit(self.bwd_ops, "UZ_OP_CONV_BWD_DATA",
                       p=[gy, self.P(wkey, wextra), self.gview(x), ("scratch", "wgrad"), self.amax_in(gy), ("amax", 0), packed]
                         + ([x, fold["part"], fold["amax"]] if fold else []),
                       i=[cout, gy.Ctot, cin, x.Ctot, x.N, x.H, x.W, ks, acc] + ([x.Ctot, 1] if fold else []), n=ws2)
        if decoupled:
            self._newgroup()
            emit_wgrad()

    def _packed(self, wkey, cin, cout, x, dgrad, vol=False):
        """Reference to the pre-packed split-fp16 weight image of this layer and direction, or None when the call packs its
        own (layers off the split path, extra tapes).  All images of a tape are packed by ONE launch at its head
        (uz_conv_pack_weights) instead of one small launch in front of every convolution."""
        # a Conv3d runs as a 2-D convolution whose contraction side has 3 C channels (the depth window): the image is packed
        # straight from the [Cout][Cin][3][3][3] parameter (no permutation pass)
        gcin, gcout = (cin, 3 * cout) if (vol and dgrad) else ((3 * cin, cout) if vol else (cin, cout))
        if os.environ.get("UZ_PREPACK", "1") != "1" or self.L.uz_conv_route(1 if dgrad else 0, gcin, gcout, x.N, x.H, x.W, 3) != 1:
            return None
        if not dgrad and self.target is not self.fwd_ops and self.target is not self.bwd_ops:
            return None                                        # extra (decode) tapes: the call packs its own image
        which = "bwd" if dgrad else "fwd"
        if which == "fwd" and self.target is self.bwd_ops and wkey not in self._packs["fwd"]:
            return None                                        # (a forward convolution that only exists in the backward tape)
        lst = self._packs[which]
        if wkey not in lst:
            lst[wkey] = dict(idx=len(lst), cin=cin, cout=cout, W=x.W, dgrad=int(dgrad), vol=int(vol),
                             mc=gcin if dgrad else gcout, kc=gcout if dgrad else gcin,
                             bytes=self.L.uz_conv_packed_bytes(gcin, gcout, x.W, int(dgrad)),
                             rows=self.L.uz_conv_pack_rows(gcin, gcout, x.W, int(dgrad)), cot=self.L.uz_conv_pack_cot(gcin, gcout, x.W, int(dgrad)))
        e = lst[wkey]
        assert (e["cin"], e["cout"], e["W"]) == (cin, cout, x.W), f"{wkey}: one weight, two convolution shapes"
        return ("packw", which, wkey)

    def _window_input(self, x, wkey):
        """The depth window of a 3x3x3 convolution addresses slices d-1, d, d+1 of its input as 3 C consecutive channels, which
        needs the view to span its buffer (C == Ctot).  A channel slice (one half of a reversible block's tensor, the image
        channels of the posterior's input) is first copied into a buffer of its own - pooled scratch inside reversible
        blocks, where the backward tape recomputes it together with everything else."""
        if x.C == x.Ctot:
            return x
        ctx = self.__dict__.get("_rev_ctx")
        if ctx is not None:
            xc = self._pool((ctx, "win" + ("F" if ".f_block." in wkey else "G")), x.C, x.H, x.W)
            for b0 in (0, x.N + 1):                          # pooled scratch is shared between shapes: re-zero the depth padding
                self._emit(self.target, "UZ_OP_MEMSET", p=[View(xc.buf, 0, x.C, b0, 1)], n=4 * x.C * x.H * x.W)
        else:
            xc = self.vol(wkey + ":win", x.C, x.N, x.H, x.W, requires_grad=False)
        self.add_views(x, None, xc)
        self.__dict__.setdefault("_win_of", {})[wkey] = xc
        return xc

    def _decouple_wgrad(self, x, ks, name=""):
        """Weight gradient as a scheduling group of its own (needs dy in a real buffer instead of the lane scratch): on the
        latency-bound planes (decouple_wgrad_px) and in the sub-networks named by decouple_wgrad_prefixes - chains whose backward
        has no independent partner chain for the second lane, so the weight gradient of layer L runs beside BatchNorm backward +
        data gradient of layer L - 1 (PHiSeg's likelihood: +1 %; its posterior / prior nets: -0.5 ... -1 %, they already pair up)."""
        if x.nb is not None or ks != 3 or self.n_lanes <= 1 or self.__dict__.get("_in_rev", False):
            return False
        lim = int(os.environ.get("UZ_DECOUPLE_WGRAD", str(self.decouple_wgrad_px)))
        pref = os.environ.get("UZ_DECOUPLE_PREFIX")
        pref = tuple(q for q in pref.split(",") if q) if pref is not None else tuple(self.__dict__.get("decouple_wgrad_prefixes", ()))
        return x.N * x.H * x.W <= lim or any(name.startswith(q) for q in pref)

    def _gy_scratch(self, like):
        self.scratch["gy"] = max(self.scratch["gy"], like.N * like.C * like.H * like.W)
        return ("gyview", like.C)

    def conv_bn_relu(self, x, cprefix, bprefix, out=None, name=None, relu=True, ybuf=None, save=None, a_grad=None):
        """Conv2D unit: Conv2d -> BatchNorm2d(eps=1e-3, momentum=0.01) -> ReLU (torchlayers.py:7-29).
        ybuf / save: caller-owned buffers for the pre-normalisation output and the saved statistics (reversible blocks share
        them); a_grad: view holding the gradient w.r.t. the unit's output instead of grad(a) (no copy)."""
        wkey, bkey = cprefix + ".weight", cprefix + ".bias"
        cout, ks = self.ptab.shape[wkey][0], self.ptab.shape[wkey][2]
        name = name or cprefix
        new = (lambda nm: self.vol(nm, cout, x.N, x.H, x.W)) if x.nb is not None else (lambda nm: self.buf(nm, cout, x.H, x.W))
        y = ybuf if ybuf is not None else new(name + ":y")
        a = out if out is not None else new(name + ":a")
        assert a.C == cout and a.H == x.H and a.W == x.W
        # (4 C floats: mean, rstd and - where the statistics come from the convolution's partials - alpha, beta' for a data gradient
        # that folds this unit's backward reduction, see _round4_passes)
        save = save if save is not None else self.vec(name + ":bnsave", 4 * cout)
        # BatchNorm statistics in the convolution's epilogue (torchlayers.py:18-21: every Conv2d of a unit feeds a BatchNorm2d):
        # where the forward kernel supports it, it leaves per-tile {sum, sum of squares, max, max(-y)} partials and the
        # BatchNorm finalises them instead of streaming y a second time
        npart = 0
        # (not where the BatchNorm forward is ONE launch that forms the statistics from the channel's batch in registers anyway:
        #  the small path, and the mid path up to uz_bn_fwd_fused_limit)
        if self.bn_training and os.environ.get("UZ_BN_FUSE_STATS", "1") == "1" and x.N * x.H * x.W > self.L.uz_bn_fwd_fused_limit(x.H, x.W) and \
                (x.nb is None or (ks == 3 and os.environ.get("UZ_BN_FUSE_STATS_VOL", "1") == "1")):
            # (a volume's Conv3d is the 2-D kernel over its slices with the depth window as 3 Cin input channels: same epilogue)
            npart = self.L.uz_conv_bn_partials(x.C if x.nb is None else 3 * x.C, cout, x.N, x.H, x.W, ks)
        bnpart = self.vec(name + ":bnpart", 4 * cout * npart) if npart else None
        # Small planes (the 8x8 ... 2x2 levels: a unit is conv -> split-K reduce -> BatchNorm, 5 - 30 us each, on the step's critical
        # chains): the convolution stops after its main kernel and the one-workgroup-per-channel BatchNorm adds the slabs itself -
        # one launch and one pass over y less per unit, bit-identical values (include/uz_api.h: uz_conv_fwd_slabs)
        nslab = 0
        if x.nb is None and x.N * x.H * x.W <= 4096 and os.environ.get("UZ_BN_FOLD_REDUCE", "1") == "1":
            nslab = self.L.uz_conv_splitk_parts(x.C, cout, x.N, x.H, x.W, ks)
            nslab = nslab if nslab > 1 else 0
        self._conv_fwd(x, wkey, bkey, y, ks, 0, bn_partials=bnpart, slabs_only=nslab > 0)
        bnws = self.L.uz_bn_workspace(cout, x.N, x.H, x.W)
        self.scratch["bn"] = max(self.scratch["bn"], bnws)
        gam, bet = bprefix + ".weight", bprefix + ".bias"
        # unit record for the round-4 passes (finalize -> _round4_passes); units inside reversible sequences (shared scratch,
        # recomputation inside the backward tape) and volume units stay as they are
        plain = ybuf is None and a_grad is None and x.nb is None and self.__dict__.get("_rev_ctx") is None and self.target is self.fwd_ops
        unit = dict(name=name, y=y, a=a, save=save, relu=int(relu), npart=npart, C=cout, N=x.N, H=x.H, W=x.W) if plain else None
        if unit is not None:
            self.__dict__.setdefault("_units", []).append(unit)
        op_fwd = self._emit(self.target, "UZ_OP_BN_RELU_FWD",
                   p=[y, self.P(gam), self.P(bet), self.B(bprefix + ".running_mean"), self.B(bprefix + ".running_var"),
                      save, a, ("scratch", "bn"), self.amax_out(a), bnpart,
                      ("scratch", "wgrad") if nslab else None, (self.P(bkey) if bkey else None) if nslab else None],
                   i=[cout, y.Ctot, a.Ctot, x.N, x.H, x.W, int(self.bn_training), int(relu), npart, nslab], f=[BN_EPS, BN_MOMENTUM])
        if unit is not None:
            unit["bn_fwd"] = op_fwd

        def bwd():
            if a_grad is None and not self._has_grad(a):
                return
            if not self.bn_training:
                raise RuntimeError("backward through eval-mode BatchNorm is not supported")
            gy = self._gy_scratch(y)
            gyv = _ScratchView(y.N, cout, y.H, y.W, amax=self._new_amax(bwd=True))
            if self._decouple_wgrad(x, ks, cprefix) and ybuf is None and a_grad is None:
                gyv.view = self.buf(name + ":dy", cout, y.H, y.W, N=y.N, requires_grad=False)
                gy = gyv
            if x.nb is not None and ks == 3:
                # volume: dy sits one slice into the scratch, between two zeroed border slices (the data gradient reads it
                # through a depth window)
                # ... in a scratch region of its own per (slices, slice size): the arena is zero-initialised and only the interior
                # slices are ever written, so the two border slices stay zero without a memset per layer and step
                sl = cout * y.H * y.W
                gyv.zkey = (y.N, sl)
                self._gyz[gyv.zkey] = (y.N + 2) * sl
                gy, gyv.off = ("gyvol", gyv.zkey), sl
            ga = a_grad if a_grad is not None else self.gview(a)
            # conv-bias gradient (the sum of dy: analytically zero behind a training-mode BatchNorm, the reference returns its
            # rounding noise and so do we): outside data-parallel runs the large-plane units leave their per-workgroup sums in rows of
            # their own and ONE table-driven launch at the end of the tape adds them - a summation launch less in every unit's chain
            dbrows = self.L.uz_bn_bwd_dbias_rows(x.N, x.H, x.W) if (plain and self._tables_ok() and os.environ.get("UZ_DBIAS_TABLE", "1") == "1") else 0
            dbpart = self.vec(name + ":dbrows", 2 * dbrows * cout) if dbrows else None
            op_bwd = self._emit(self.bwd_ops, "UZ_OP_BN_RELU_BWD",
                                p=[ga, y, self.P(gam), self.P(bet), save, gy, self.G(gam), self.G(bet), dbpart if dbrows else self.G(bkey), ("scratch", "bn"),
                                   ("amax", gyv.amax, "acc")],
                                i=[ga.Ctot, cout, y.Ctot, cout, x.N, x.H, x.W, int(relu), 0, 0, 1 if dbrows else 0])
            if dbrows:
                self._dbias_jobs.append((dbpart, bkey, dbrows, cout, 1))
            rec = {}
            self._conv_bwd(x, wkey, gyv, ks, rec=rec)
            if x.nb is not None and ks == 3 and ybuf is None and a_grad is None and self.__dict__.get("_rev_ctx") is None:
                # volume unit: its three backward ops share the zero-bordered dy scratch (see _b16_pass)
                self.__dict__.setdefault("_vol_units", []).append(dict(bn_bwd=op_bwd, gyv=gyv, wgrad=rec.get("wgrad"), dgrad=rec.get("dgrad")))
            if unit is not None:
                unit.update(bn_bwd=op_bwd, ga=ga, ks=ks, cin=x.C, wgrad=rec.get("wgrad"), dgrad=rec.get("dgrad"))
        self._push_bwd(bwd)
        return a

    # ------------------------------------------------------------------ reversible blocks
    def _pool(self, key, C, H, W, vec=False):
        """Scratch activations of the reversible blocks: one buffer per (branch, role, shape), re-used by every block that
        needs that role - nothing inside a reversible sequence is kept for the backward pass, it is recomputed."""
        pool = self.__dict__.setdefault("_rev_pool", {})
        slots = self.__dict__.setdefault("_rev_slots", {})
        k = (key, C, H, W)
        if k not in pool:
            nm = "revpool:" + "/".join(map(str, k))
            D = self.__dict__.get("_vol_depth")
            v = self.vec(nm, C) if vec else (self.vol(nm, C, D, H, W) if D else self.buf(nm, C, H, W))
            # all shapes of one (branch, role) share ONE region sized for the largest of them: a reversible sequence's scratch is
            # dead as soon as the next sequence of the branch starts (forward) or has been back-propagated (backward)
            # (pooled volumes therefore have no guaranteed zero border slices: _window_input clears those of the one kind of
            # pooled volume that is read through a depth window)
            slot = slots.setdefault(key[:2], {"floats": 0, "off": None})
            slot["floats"] = max(slot["floats"], v.buf.numel)
            v.buf.alias = slot
            pool[k] = v
        return pool[k]

    def add_views(self, a, b, y, alpha=1.0, accumulate=0, target=None):
        self._emit(target if target is not None else self.target, "UZ_OP_ADD_VIEWS",
                   p=[a, b, y, self.amax_in(a), self.amax_in(b) if b is not None else None,
                      self.amax_out(y) if (self.amax_in(a) is not None and (b is None or self.amax_in(b) is not None)) else None],
                   i=[a.Ctot, b.Ctot if b is not None else 0, y.Ctot, y.C, y.N, y.H, y.W, accumulate], f=[alpha])

    def rev_sequence(self, x, prefix, cout, depth, unit, out=None):
        """ReversibleSequence (torchlayers.py:55-82): an optional 1x1 Conv2D unit to `cout` channels (`inital_conv`), then
        `depth` additive-coupling blocks of revtorch (revtorch==0.2.0, requirements.txt:38, not vendored - restated from
        its published algorithm): split the channels into halves (x1, x2);
            y1 = x1 + F(x2),  y2 = x2 + G(y1),      F, G = Conv2D(c/2 -> c/2, 3x3) units (conv + BN + ReLU)
        and in the backward pass recompute the inputs from the outputs instead of storing them:
            x2 = y2 - G(y1),  x1 = y1 - F(x2),      then back-propagate through G and F on the recomputed tensors.
        The blocks' activations live in pooled scratch buffers (forward and backward share them); only the sequence's
        output is kept - that is the memory saving the reference advertises (README.md:4-5).  Like revtorch, the
        recomputation re-runs F and G in training mode, so their BatchNorm running statistics receive a second momentum
        update per step.  `unit(plan, x, prefix, **kw)` emits one Conv2D unit."""
        branch = prefix.split(".", 1)[0]
        N, H, W, h = x.N, x.H, x.W, cout // 2
        assert cout % 2 == 0
        self._vol_depth = x.N if x.nb is not None else None      # volumes: pooled scratch is volume-shaped (and keyed by depth below)
        if x.nb is not None:
            branch = (branch, x.N)
        seq_out = self.vol(prefix + ":revout", cout, x.N, H, W) if x.nb is not None else self.buf(prefix + ":revout", cout, H, W)
        widened = x.C != cout
        if widened:
            # the 1x1 unit's output is only the first block's input: it is recomputed in backward like every other block
            # input, so it lives in pooled scratch, and its gradient IS the dX the blocks form in the sequence's gradient buffer
            x = unit(self, x, prefix + ".inital_conv", out=self._pool((branch, "xin"), cout, H, W), a_grad=self.gview(seq_out))
        record = self._record_bwd
        blocks = []
        cur = x
        for i in range(depth):
            y = seq_out if i == depth - 1 else self._pool((branch, "x%d" % (i & 1)), cout, H, W)
            bp = f"{prefix}.sequence.reversible_blocks.{i}"
            self._rev_block(cur, y, bp, unit, h, branch, forward=True)
            blocks.append((cur, y, bp))
            cur = y
        if out is not None:                                  # consumer-owned slice of a concat buffer: strided copy
            self._newgroup()
            self.add_views(seq_out, None, out)

        def bwd():
            gsrc = seq_out
            if out is not None:
                if not self._has_grad(out):
                    return
                acc = self._claim(seq_out)
                self.add_views(self.gview(out), None, self.gview(seq_out), accumulate=acc, target=self.bwd_ops)
            if not self._has_grad(seq_out):
                return
            ycur = seq_out
            self._vol_depth = seq_out.N if seq_out.nb is not None else None
            for i in reversed(range(depth)):
                xin, _, bp = blocks[i]
                xrec = self._pool((branch, "x%d" % (i & 1)), cout, H, W)       # same ping-pong pair as the forward pass
                self._ginit.pop(xrec.buf, None)
                xrec.buf.gbuf = self.gview(seq_out).buf           # dX is formed in place in the sequence's gradient buffer
                self._ginit[xrec.buf] = [(0, cout)]
                self._rev_block(xrec, ycur, bp, unit, h, branch, forward=False)
                ycur = xrec
            if x.buf.requires_grad and not widened:
                acc = self._claim(x)
                self._newgroup()
                self.add_views(self.gview(seq_out), None, self.gview(x), accumulate=acc, target=self.bwd_ops)
        if record:
            self._push_bwd(bwd)
        return out if out is not None else seq_out

    def _rev_block(self, xv, yv, bp, unit, h, branch, forward):
        """One additive-coupling block.  forward=True: y from x (forward tape).  forward=False (inside the backward tape):
        x recomputed from y into `xv`, then the gradients of F, G and dX (in place in the shared gradient buffer)."""
        N, H, W = xv.N, xv.H, xv.W
        self._vol_depth = xv.N if xv.nb is not None else None
        self._rev_ctx = branch
        x1, x2, y1, y2 = xv.slice(0, h), xv.slice(h, h), yv.slice(0, h), yv.slice(h, h)
        bufs = {}
        for tag in ("F", "G"):
            ybuf, abuf = self._pool((branch, tag + "y"), h, H, W), self._pool((branch, tag + "a"), h, H, W)
            save = self._pool((branch, tag + "s"), 4 * h, 1, 1, vec=True)
            for b in (ybuf, abuf):
                self._ginit.pop(b.buf, None)
            bufs[tag] = (ybuf, abuf, save)
        if forward:
            rec, self._record_bwd = self._record_bwd, False      # nothing of the block is recorded: backward recomputes it
            tF = unit(self, x2, bp + ".f_block.0", out=bufs["F"][1], ybuf=bufs["F"][0], save=bufs["F"][2])
            self._newgroup()
            self.add_views(x1, tF, y1)
            tG = unit(self, y1, bp + ".g_block.0", out=bufs["G"][1], ybuf=bufs["G"][0], save=bufs["G"][2])
            self._newgroup()
            self.add_views(x2, tG, y2)
            self._record_bwd = rec
            self._rev_ctx = None
            return
        gy = self.gview(yv)
        gy1, gy2 = gy.slice(0, h), gy.slice(h, h)
        tgt, rec, saved_bwd = self.target, self._record_bwd, self._bwd
        self.target, self._record_bwd, self._bwd = self.bwd_ops, True, []
        try:
            # x2 = y2 - G(y1); gradients through G: dy1 += G'(dy2) (in place in the gradient buffer)
            self._newgroup()
            tG = unit(self, y1, bp + ".g_block.0", out=bufs["G"][1], ybuf=bufs["G"][0], save=bufs["G"][2], a_grad=gy2, recompute=True)
            self._newgroup()
            self.add_views(y2, tG, x2, alpha=-1.0)
            g_closures, self._bwd = self._bwd, []
            # x1 = y1 - F(x2); gradients through F: dx2 = dy2 + F'(dy1_total) (in place), dx1 = dy1_total (already there)
            self._newgroup()
            tF = unit(self, x2, bp + ".f_block.0", out=bufs["F"][1], ybuf=bufs["F"][0], save=bufs["F"][2], a_grad=gy1, recompute=True)
            self._newgroup()
            self.add_views(y1, tF, x1, alpha=-1.0)
            f_closures = self._bwd
            for fn in reversed(g_closures):
                self._newgroup()
                fn()
            for fn in reversed(f_closures):
                self._newgroup()
                fn()
        finally:
            self.target, self._record_bwd, self._bwd = tgt, rec, saved_bwd
            self._rev_ctx = None

    def conv_relu(self, x, prefix, out=None, name=None):
        """nn.Conv2d(3, pad 1) + nn.ReLU(inplace=True) of the vanilla U-Net blocks (unet.py:25-30)."""
        wkey, bkey = prefix + ".weight", prefix + ".bias"
        cout, _, ks, _ = self.ptab.shape[wkey]
        a = out if out is not None else self.buf((name or prefix) + ":a", cout, x.H, x.W)
        self._conv_fwd(x, wkey, bkey, a, ks, 1)
        self.scratch["bn"] = max(self.scratch["bn"], self.L.uz_bn_workspace(cout, x.N, x.H, x.W))
        if a.buf.relu is None:
            a.buf.relu = {}
        state = a.buf.relu[(a.c0, a.C)] = {}

        def bwd():
            if not self._has_grad(a):
                return
            fold = state.get("fold")
            if fold is not None:
                # the data gradient that wrote grad(a) last already applied this unit's ReLU mask (see _conv_bwd)
                if self._nclaims.get(a.buf, 0) != fold["claims"]:
                    raise RuntimeError(f"{prefix}: a later writer of grad({a.buf.name}) follows the data gradient that folded the ReLU "
                                       "backward - set UZ_FOLD_RELU_BWD=0 for this model")
                if not self._tables_ok() or os.environ.get("UZ_DBIAS_TABLE", "1") != "1":
                    self._emit(self.bwd_ops, "UZ_OP_CHAN_SUM_PARTIALS", p=[fold["part"], self.G(bkey)], i=[fold["npart"], cout, fold.get("dbl", 0)])
                else:
                    self._dbias_jobs.append((fold["part"], bkey, fold["npart"], cout, fold.get("dbl", 0)))
                self._conv_bwd(x, wkey, self.gview(a), ks)
                return
            gy = self._gy_scratch(a)
            gyv = _ScratchView(a.N, cout, a.H, a.W, amax=self._new_amax(bwd=True))
            self._emit(self.bwd_ops, "UZ_OP_RELU_BWD",
                       p=[self.gview(a), a, gy, self.G(bkey), ("scratch", "bn"), ("amax", gyv.amax, "acc")],
                       i=[a.Ctot, cout, a.Ctot, cout, x.N, x.H, x.W])
            self._conv_bwd(x, wkey, gyv, ks)
        self._push_bwd(bwd)
        return a

    def conv_bare(self, x, prefix, out=None, name=None, rows=None):
        """Plain nn.Conv2d (1x1 heads: phiseg.py:95-96,281-284; unet.py:122; probabilistic_unet.py:95).
        rows=(r0, n): use only output rows [r0, r0+n) of the parameter (mu / log-sigma halves)."""
        wkey, bkey = prefix + ".weight", prefix + ".bias"
        cout, ks = self.ptab.shape[wkey][0], self.ptab.shape[wkey][2]
        r0 = 0
        if rows is not None:
            r0, cout = rows
        y = out if out is not None else (self.vol((name or prefix) + ":y", cout, x.N, x.H, x.W) if x.nb is not None
                                         else self.buf((name or prefix) + ":y", cout, x.H, x.W))
        self._conv_fwd(x, wkey, bkey, y, ks, 0, wrow0=r0)

        def bwd():
            if not self._has_grad(y):
                return
            self._conv_bwd(x, wkey, self.gview(y), ks, db_key=bkey, wrow0=r0)
        self._push_bwd(bwd)
        return y

    # ------------------------------------------------------------------ resampling
    def _resample(self, fcode, bcode, x, y, extra_i=()):
        self._newgroup()
        pp = [x, y]
        if fcode in ("UZ_OP_AVGPOOL_FWD", "UZ_OP_BILINEAR_FWD"):
            # pooling and interpolation are convex combinations: the output inherits the input's magnitude bound
            xin = self.amax_in(x)
            pp += [xin, self.amax_out(y) if xin is not None else None]
        self._emit(self.target, fcode, p=pp, i=[x.C, x.Ctot, y.Ctot, x.N, x.H, x.W, *extra_i])

        def bwd():
            if not self._has_grad(y) or not x.buf.requires_grad:
                return
            acc = self._claim(x)
            # x is the output of a Conv -> ReLU unit and this pooling / interpolation backward is the last writer of its gradient
            # (the third unit of every vanilla U-Net block): fold that unit's ReLU backward in, like _conv_bwd does
            fold = None
            st = (x.buf.relu or {}).get((x.c0, x.C)) if x.nb is None else None
            if st is not None and bcode in ("UZ_OP_AVGPOOL_BWD", "UZ_OP_BILINEAR_BWD") and not self.__dict__.get("_in_rev", False) \
                    and os.environ.get("UZ_FOLD_RELU_BWD", "1") == "1":
                rows = self.L.uz_resample_bwd_relu_rows(0 if bcode == "UZ_OP_AVGPOOL_BWD" else 1, x.C, x.N, x.H, x.W)
                fold = st["fold"] = dict(part=self.vec(x.buf.name + f":dbpart{x.c0}", 2 * rows * x.C), npart=rows, dbl=1,
                                         amax=self.amax_out(self.gview(x)), claims=self._nclaims[x.buf])
            self._emit(self.bwd_ops, bcode, p=[self.gview(y), self.gview(x)] + ([x, fold["part"], fold["amax"]] if fold else []),
                       i=[x.C, y.Ctot, x.Ctot, x.N, x.H, x.W, *extra_i, acc] + ([x.Ctot] if fold else []))
        self._push_bwd(bwd)
        return y

    def avgpool(self, x, name):
        """nn.AvgPool2d(2, 2, ceil_mode=True) (phiseg.py:23, unet.py:22)."""
        y = self.buf(name, x.C, (x.H + 1) // 2, (x.W + 1) // 2, requires_grad=x.buf.requires_grad)
        return self._resample("UZ_OP_AVGPOOL_FWD", "UZ_OP_AVGPOOL_BWD", x, y)

    def bilinear(self, x, align_corners, name=None, out=None):
        y = out if out is not None else self.buf(name, x.C, 2 * x.H, 2 * x.W, requires_grad=x.buf.requires_grad)
        return self._resample("UZ_OP_BILINEAR_FWD", "UZ_OP_BILINEAR_BWD", x, y, (int(align_corners),))

    def nearest(self, x, factor, name):
        y = self.buf(name, x.C, x.H * factor, x.W * factor, requires_grad=x.buf.requires_grad)
        if factor == 1:
            pass
        return self._resample("UZ_OP_NEAREST_FWD", "UZ_OP_NEAREST_BWD", x, y, (factor,))

    def avgpool3d(self, x, name):
        """nn.AvgPool3d(2, 2, ceil_mode=True) (phiseg3D.py:101) on a volume."""
        D, H, W = x.N, x.H, x.W
        y = self.vol(name, x.C, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2, requires_grad=x.buf.requires_grad)
        self._newgroup()
        self._emit(self.target, "UZ_OP_AVGPOOL3D_FWD", p=[x, y], i=[x.C, x.Ctot, y.Ctot, D, H, W])
        if self.amax_in(x) is not None:                       # a mean of inputs: the input's bound holds (forwarded by a 1-op copy)
            self.add_views_bound(x, y)

        def bwd():
            if not self._has_grad(y) or not x.buf.requires_grad:
                return
            acc = self._claim(x)
            self._emit(self.bwd_ops, "UZ_OP_AVGPOOL3D_BWD", p=[self.gview(y), self.gview(x)], i=[x.C, y.Ctot, x.Ctot, D, H, W, acc])
        self._push_bwd(bwd)
        return y

    def add_views_bound(self, x, y):
        """Make y's magnitude-bound slot inherit x's (one lane of a tiny launch)."""
        self._emit(self.target, "UZ_OP_ABSMAX_COPY", p=[self.amax_in(x), self.amax_out(y)])

    def trilinear(self, x, name=None, out=None):
        """F.interpolate(mode='trilinear', scale_factor=2, align_corners=True) (phiseg3D.py:146,306,376): the 2-D bilinear kernel
        per slice, then linear interpolation along the depth (csrc/vol.hip) - the interpolation is separable."""
        D, H, W = x.N, x.H, x.W
        mid = self.vol((name or "tri") + ":inplane", x.C, D, 2 * H, 2 * W, requires_grad=x.buf.requires_grad)
        self.bilinear(x, True, out=mid)
        y = out if out is not None else self.vol(name, x.C, 2 * D, 2 * H, 2 * W, requires_grad=x.buf.requires_grad)
        self._newgroup()
        self._emit(self.target, "UZ_OP_DEPTH_LERP_FWD", p=[mid, y], i=[x.C, mid.Ctot, y.Ctot, D, 2 * H, 2 * W])
        if self.amax_in(mid) is not None:
            self.add_views_bound(mid, y)

        def bwd():
            if not self._has_grad(y) or not x.buf.requires_grad:
                return
            acc = self._claim(mid)
            self._emit(self.bwd_ops, "UZ_OP_DEPTH_LERP_BWD", p=[self.gview(y), self.gview(mid)], i=[x.C, y.Ctot, mid.Ctot, D, 2 * H, 2 * W, acc])
        self._push_bwd(bwd)
        return y

    def nearest3d(self, x, f, fz, name):
        y = self.vol(name, x.C, x.N * fz, x.H * f, x.W * f, requires_grad=x.buf.requires_grad)
        self._newgroup()
        self._emit(self.target, "UZ_OP_NEAREST3D_FWD", p=[x, y], i=[x.C, x.Ctot, y.Ctot, x.N, x.H, x.W, f, fz])

        def bwd():
            if not self._has_grad(y) or not x.buf.requires_grad:
                return
            acc = self._claim(x)
            self._emit(self.bwd_ops, "UZ_OP_NEAREST3D_BWD", p=[self.gview(y), self.gview(x)], i=[x.C, y.Ctot, x.Ctot, x.N, x.H, x.W, f, fz, acc])
        self._push_bwd(bwd)
        return y

    def spatial_mean(self, x, name):
        y = self.buf(name, x.C, 1, 1)
        self._newgroup()
        self._emit(self.target, "UZ_OP_SPATIAL_MEAN_FWD", p=[x, y], i=[x.C, x.Ctot, x.N, x.H, x.W])

        def bwd():
            if not self._has_grad(y):
                return
            acc = self._claim(x)
            self._emit(self.bwd_ops, "UZ_OP_SPATIAL_MEAN_BWD", p=[self.gview(y), self.gview(x)],
                       i=[x.C, x.Ctot, x.N, x.H, x.W, acc])
        self._push_bwd(bwd)
        return y

    def bcast_channels(self, z, out):
        """Fcomb tiling of z (N, L) over the spatial axes into channels of `out` (probabilistic_unet.py:190-197)."""
        L = z.C
        self._newgroup()
        self._emit(self.target, "UZ_OP_BCAST_CHANNELS", p=[z, out], i=[L, out.Ctot, out.N, out.H, out.W])
        if self._record_bwd:
            # another tape (decode: sample() / reconstruct()) may overwrite these channels between loss() and
            # backward(), which the reference allows; the backward tape therefore re-tiles z before it reads them
            self._bwd_head.append(lambda: self._emit(self.bwd_ops, "UZ_OP_BCAST_CHANNELS", p=[z, out],
                                                     i=[L, out.Ctot, out.N, out.H, out.W]))

        def bwd():
            if not self._has_grad(out) or not z.buf.requires_grad:
                return
            assert self._claim(z) == 0
            self._emit(self.bwd_ops, "UZ_OP_BCAST_CHANNELS_BWD", p=[self.gview(out), self.gview(z)],
                       i=[out.Ctot, L, out.N, out.H, out.W])
        self._push_bwd(bwd)

    # ------------------------------------------------------------------ inputs / latents / losses
    def posterior_input(self, patch, mask, nlabels, name):
        out = self.buf(name, patch.C + nlabels, patch.H, patch.W, requires_grad=False)
        self._newgroup()
        self._emit(self.target, "UZ_OP_POSTERIOR_INPUT", p=[patch, mask, out], i=[patch.C, nlabels, patch.N, patch.H, patch.W])
        return out

    def latent(self, mu, pre, eps, name, want_z=True, act=0):
        assert mu.contiguous and pre.contiguous
        new = (lambda nm: self.vol(nm, mu.C, mu.N, mu.H, mu.W)) if mu.nb is not None else (lambda nm: self.buf(nm, mu.C, mu.H, mu.W, N=mu.N))
        sigma = new(name + ":sigma")
        z = new(name + ":z") if want_z else None
        lat = Latent(mu, pre, sigma, z, eps, act)
        self._newgroup()
        self._emit(self.target, "UZ_OP_LATENT_FWD", p=[mu, pre, eps, sigma, z], i=[act], n=mu.numel)

        def bwd():
            dz = self.gview(z) if (z is not None and self._has_grad(z)) else None
            if dz is None and lat.kl_dmu is None:
                return
            assert self._claim(mu) == 0 and self._claim(pre) == 0
            self._emit(self.bwd_ops, "UZ_OP_LATENT_BWD",
                       p=[lat.kl_dmu, lat.kl_dsigma, dz, eps, sigma, self.gview(mu), self.gview(pre)], i=[act], n=mu.numel)
        self._push_bwd(bwd)
        return lat

    def latent_heads(self, h, mu_prefix, sigma_prefix, eps, name, want_z=True, act=0):
        """The tail of a SampleZBlock (phiseg.py:95-105): mu = mu_conv(h), pre = sigma_conv(h), sigma = softplus(pre), z = mu + sigma eps.
        On 2-D fp32 planes the three forward launches become ONE (h read once) and the six launches of the two heads' backward THREE
        (one data gradient that writes dh once, one weight-gradient pair that reads h once) - csrc/conv1x1_small.hip, bit-identical to the
        separate ops (UZ_FUSE_HEADS=0 keeps those).  Returns the Latent like latent()."""
        wm, bm, wsg, bsg = mu_prefix + ".weight", mu_prefix + ".bias", sigma_prefix + ".weight", sigma_prefix + ".bias"
        L, cin = self.ptab.shape[wm][0], h.C
        fuse = (os.environ.get("UZ_FUSE_HEADS", "1") == "1" and h.nb is None and self.ptab.shape[wm][2:] == (1, 1) and self.ptab.shape[wsg] == self.ptab.shape[wm]
                and len(self.ptab.shape[wm]) == 4 and bool(self.L.uz_latent_heads_ok(cin, L)) and not self.__dict__.get("_in_rev", False))
        if not fuse:
            mu = self.conv_bare(h, mu_prefix)
            ps = self.conv_bare(h, sigma_prefix)
            return self.latent(mu, ps, eps, name, want_z=want_z, act=act)
        mu = self.buf(mu_prefix + ":y", L, h.H, h.W)
        pre = self.buf(sigma_prefix + ":y", L, h.H, h.W)
        sigma = self.buf(name + ":sigma", L, h.H, h.W)
        z = self.buf(name + ":z", L, h.H, h.W) if want_z else None
        lat = Latent(mu, pre, sigma, z, eps, act)
        self._newgroup()
        self._emit(self.target, "UZ_OP_LATENT_HEADS_FWD", p=[h, self.P(wm), self.P(bm), self.P(wsg), self.P(bsg), eps, mu, pre, sigma, z],
                   i=[cin, h.Ctot, L, h.N, h.H, h.W, act])

        def bwd():
            dz = self.gview(z) if (z is not None and self._has_grad(z)) else None
            if dz is None and lat.kl_dmu is None:
                return
            assert self._claim(mu) == 0 and self._claim(pre) == 0
            self._emit(self.bwd_ops, "UZ_OP_LATENT_BWD",
                       p=[lat.kl_dmu, lat.kl_dsigma, dz, eps, sigma, self.gview(mu), self.gview(pre)], i=[act], n=mu.numel)
            # head a = the sigma head: its separate backward ran first (reverse order of the forward), so its rows are added first.
            # The data gradient - what the unit below waits for - goes first; the weight gradients are a scheduling group of their own.
            if h.buf.requires_grad:
                acc = self._claim(h)
                self._emit(self.bwd_ops, "UZ_OP_LATENT_HEADS_BWD_DATA", p=[self.gview(pre), self.gview(mu), self.P(wsg), self.P(wm), self.gview(h)],
                           i=[L, cin, h.Ctot, h.N, h.H, h.W, acc])
            ws = self.L.uz_latent_heads_bwd_weight_workspace(cin, L, h.N, h.H, h.W)
            self.scratch["wgrad"] = max(self.scratch["wgrad"], ws)
            self._newgroup()
            self._emit(self.bwd_ops, "UZ_OP_LATENT_HEADS_BWD_WEIGHT",
                       p=[h, self.gview(pre), self.gview(mu), self.G(wsg), self.G(bsg), self.G(wm), self.G(bm), ("scratch", "wgrad")],
                       i=[cin, h.Ctot, L, h.N, h.H, h.W], n=ws)
        self._push_bwd(bwd)
        return lat

    def kl(self, q, p, weight, term):
        """weight * KL_two_gauss_with_diag_cov(q || p) (phiseg.py:436-479)."""
        n, per = q.mu.N, q.mu.C * q.mu.H * q.mu.W
        if q.mu.nb is not None:                                 # volume: ONE sample whose depth slices are stored as the batch
            n, per = 1, q.mu.N * per
        self._newgroup()
        self.scratch["bn"] = max(self.scratch["bn"], 512)
        self._emit(self.target, "UZ_OP_KL_FWD", p=[q.mu, q.sigma, p.mu, p.sigma, term, ("scratch", "bn")], i=[n, per], f=[weight])

        def bwd():
            for lat, tag in ((q, "q"), (p, "p")):
                lat.kl_dmu = self.buf(f"kl:{tag}:{lat.mu.buf.name}:dmu", lat.mu.C, lat.mu.H, lat.mu.W, requires_grad=False)
                lat.kl_dsigma = self.buf(f"kl:{tag}:{lat.mu.buf.name}:dsigma", lat.mu.C, lat.mu.H, lat.mu.W, requires_grad=False)
            self._emit(self.bwd_ops, "UZ_OP_KL_BWD",
                       p=[q.mu, q.sigma, p.mu, p.sigma, self.loss_scale, q.kl_dmu, q.kl_dsigma, p.kl_dmu, p.kl_dsigma],
                       i=[n, per], f=[weight])
        self._push_bwd(bwd)

    def residual_ce(self, s_list, mask, terms, post_scale=None):
        """residual_multinoulli_loss over the level logits (phiseg.py:481-513); terms: vec View of L floats.
        post_scale: extra factor on the terms (1/(H*W) turns the L=1 case into nn.CrossEntropyLoss() mean,
        unet.py:159-165)."""
        L, K = len(s_list), s_list[0].C
        assert all(s.contiguous for s in s_list)
        tab = self.ptr_table(list(s_list))
        ws = self.L.uz_ce_workspace(mask.N, mask.H, mask.W, L)
        self.scratch["ce"] = max(self.scratch["ce"], ws)
        self._newgroup()
        self._emit(self.target, "UZ_OP_CE_FWD", p=[tab, mask, terms, ("scratch", "ce")], i=[L, K, mask.N, mask.H, mask.W])
        if post_scale is not None:
            self._emit(self.target, "UZ_OP_SCALE", p=[terms], f=[post_scale], n=L)

        def bwd():
            for s in s_list:
                assert self._claim(s) == 0
            gtab = self.ptr_table([self.gview(s) for s in s_list])
            scale_ref = self.loss_scale
            if post_scale is not None:
                scale_ref = self.vec("ce_scale", 1)
                self._emit(self.bwd_ops, "UZ_OP_COPY", p=[scale_ref, self.loss_scale], n=4)
                self._emit(self.bwd_ops, "UZ_OP_SCALE", p=[scale_ref], f=[post_scale], n=1)
            self._emit(self.bwd_ops, "UZ_OP_CE_BWD", p=[tab, gtab, mask, scale_ref], i=[L, K, mask.N, mask.H, mask.W])
        self._push_bwd(bwd)

    def sum_terms(self, terms, n, total):
        self._newgroup()
        self._emit(self.target, "UZ_OP_SUM_TERMS", p=[terms, total], i=[n])

    def scale_(self, v, alpha, n):
        self._newgroup()
        self._emit(self.target, "UZ_OP_SCALE", p=[v], f=[alpha], n=n)

    def ptr_table(self, refs):
        self.ptr_tables.append(refs)
        return ("ptrtab", len(self.ptr_tables) - 1)

    # ------------------------------------------------------------------ round-4 passes over the emitted ops
    # p[] slots through which an op may WRITE / READ a buffer that is kept in split storage (csrc/split_f16.h: one word per element
    # holding the two fp16 pieces of the scaled value - what the split-fp16 convolutions otherwise form in their staging)
    _PACK_WRITERS = {"UZ_OP_BN_RELU_FWD": (6, 8), "UZ_OP_BILINEAR_FWD": (1, 3), "UZ_OP_AVGPOOL_FWD": (1, 3)}     # (view slot, bound slot)
    _PACK_READERS = {"UZ_OP_CONV_FWD": 0, "UZ_OP_CONV_BWD_WEIGHT": 0}

    def _round4_passes(self):
        """Plan-time rewrites on the Conv2D units (torchlayers.py:18-21) of the large planes, all decided from the emitted ops:
          1. folded backward reduction - where the ONLY writer of a unit's dA is an unsplit split-path data gradient, that launch masks
             dA with the unit's ReLU and leaves the BatchNorm-backward sums in its epilogue; the unit's backward runs a per-channel
             finalise instead of the reduction pass over dA and y (UZ_FOLD_BN_BWD=0 switches it off);
          2. dY in split storage - where both consumers of a unit's dy (its weight and data gradient) run on the split path, the
             BatchNorm-backward apply pass writes the operand pieces instead of fp32 values (UZ_PACK_DY=0);
          3. activations in split storage - a buffer written only by BatchNorm-apply / pooling / interpolation launches that know
             their bound beforehand and read only by split-path convolutions (forward + weight gradient) is kept as operand pieces;
             a concat buffer's (at most two) producers each scale from a slot of their own (UZ_PACK_ACT=0).
        Returns a summary dict (also kept as self.round4)."""
        info = dict(folded=0, dy_packed=0, act_packed=0, act_views=0)
        self.round4 = info
        units = [u for u in self.__dict__.get("_units", []) if "bn_bwd" in u or "bn_fwd" in u]
        if self.L.uz_get_conv_math() in (0, 3) or not self.bn_training:
            return info                              # fp32-only / bf16 modes have no two-piece operands
        env = lambda k: os.environ.get(k, "1") == "1"
        large = lambda u: u["N"] * u["H"] * u["W"] > 4096 and (u["H"] * u["W"]) % 4 == 0
        all_ops = [("fwd", self.fwd_ops), ("loss", self.loss_ops), ("bwd", self.bwd_ops)] + list(self.extra_ops.items())

        def touches(r, buf):
            if isinstance(r, View):
                return r.buf is buf
            if isinstance(r, _ScratchView):
                return r.view is not None and r.view.buf is buf
            if isinstance(r, tuple) and r and r[0] in ("win",):
                return r[1].buf is buf
            return False
        tabbed = {id(q.buf) for t in self.ptr_tables for q in t if isinstance(q, View)}
        # ---- 1. folded backward reduction
        # (off by default: measured on MI355X the epilogue's extra read of y is exposed - the 64-channel-tile kernel keeps ONE
        #  workgroup per CU, so its memory phase overlaps nothing: data gradient of 128 -> 128 @ 128 x 128 451 -> 573 us against
        #  269 -> 170 us for the unit's BatchNorm backward; step +-0)
        if os.environ.get("UZ_FOLD_BN_BWD", "0") == "1" and self.bwd_ops:
            for u in units:
                B = u.get("bn_bwd")
                if B is None or not large(u) or u["npart"] <= 0:
                    continue
                ga = u["ga"]
                if not isinstance(ga, View) or ga.nb is not None or id(ga.buf) in tabbed:
                    continue
                writers, readers = [], []
                for tape, ops in all_ops:
                    for o in ops:
                        wr = self._op_writes(o)
                        for j, r in enumerate(o["p"]):
                            if touches(r, ga.buf) and isinstance(r, View) and r.c0 < ga.c0 + ga.C and ga.c0 < r.c0 + r.C:
                                (writers if j in wr else readers).append((o, j, r))
                if len(writers) != 1 or len(readers) != 1 or readers[0][0] is not B:
                    continue
                W, j, r = writers[0]
                if W["code"] != "UZ_OP_CONV_BWD_DATA" or j != 2 or (r.c0, r.C, r.nb) != (ga.c0, ga.C, None) or len(W["p"]) != 7 or W["i"][8] != 0 or W["i"][7] != 3:
                    continue
                cout_w, cin_w, N, H, Wd = W["i"][0], W["i"][2], W["i"][4], W["i"][5], W["i"][6]
                rows = self.L.uz_conv_bwd_relu_partials(cin_w, cout_w, N, H, Wd, 3)
                if rows <= 0 or cin_w != u["C"]:
                    continue
                part = self.vec(u["name"] + ":bwdpart", 4 * u["C"] * rows)
                y = u["y"]
                W["p"] = W["p"][:7] + [y, part, None, u["save"]]
                W["i"] = W["i"][:9] + [y.Ctot, 2, 0, u["relu"]]
                B["p"] = B["p"][:11] + [part]
                B["i"][8] = rows
                info["folded"] += 1
        # ---- 1b. small planes: the data gradient's split-K reduce folded into the consumer's BatchNorm backward (the mirror of the
        # forward's uz_conv_fwd_slabs / uz_bn_relu_fwd_slabs): where the ONLY writer of a unit's dA is a split-K fp32 data gradient, that
        # launch leaves its partial sums in a buffer of the layer's own and the unit's one-workgroup-per-channel backward adds them
        # itself - one reduction launch less on the backward chains of the 8 x 8 ... 2 x 2 levels (UZ_BN_FOLD_DGRAD=0 switches it off)
        info["dgrad_folded"] = 0
        if env("UZ_BN_FOLD_DGRAD") and self.bwd_ops:
            for u in units:
                B = u.get("bn_bwd")
                ga = u.get("ga")
                if B is None or u["N"] * u["H"] * u["W"] > 4096 or not isinstance(ga, View) or ga.nb is not None or id(ga.buf) in tabbed:
                    continue
                writers, readers = [], []
                for tape, ops in all_ops:
                    for o in ops:
                        wr = self._op_writes(o)
                        for j, r in enumerate(o["p"]):
                            if touches(r, ga.buf) and isinstance(r, View) and r.c0 < ga.c0 + ga.C and ga.c0 < r.c0 + r.C:
                                (writers if j in wr else readers).append((o, j, r))
                if len(writers) != 1 or len(readers) != 1 or readers[0][0] is not B:
                    continue
                W, j, r = writers[0]
                if W["code"] != "UZ_OP_CONV_BWD_DATA" or j != 2 or (r.c0, r.C, r.nb) != (ga.c0, ga.C, None) or len(W["p"]) != 7 or W["i"][8] != 0:
                    continue
                cout_w, cin_w, N, H, Wd, ksw = W["i"][0], W["i"][2], W["i"][4], W["i"][5], W["i"][6], W["i"][7]
                parts = self.L.uz_conv_bwd_splitk_parts(cin_w, cout_w, N, H, Wd, ksw)
                if parts <= 1 or cin_w != u["C"]:
                    continue
                slabs = self.vec(u["name"] + ":daslabs", parts * N * cin_w * H * Wd)
                W["p"] = W["p"][:7] + [slabs]
                W["i"] = W["i"][:9] + [0, 3]
                B["p"] = (B["p"] + [None])[:11] + [slabs]
                B["i"] = (B["i"] + [0])[:11] + [parts]
                info["dgrad_folded"] += 1
        # ---- 2. dY in split storage
        if env("UZ_PACK_DY") and self.bwd_ops:
            for u in units:
                B, wg, dg = u.get("bn_bwd"), u.get("wgrad"), u.get("dgrad")
                if B is None or wg is None or not large(u) or u["ks"] != 3 or u["W"] % 4:
                    continue
                if u["N"] * u["H"] * u["W"] <= self.L.uz_bn_bwd_fused_limit(u["H"], u["W"]):
                    continue                         # one-launch backward (channel's batch on chip): no tensor-wide bound before the first write
                if self.L.uz_conv_route(2, u["cin"], u["C"], u["N"], u["H"], u["W"], 3) != 1:
                    continue
                if dg is not None and (self.L.uz_conv_route(1, u["cin"], u["C"], u["N"], u["H"], u["W"], 3) != 1 or dg["i"][10:11] == [1]):
                    continue
                B["i"][9] = 1
                B["p"] = (B["p"] + [None])[:12]
                wg["i"][10] = 1
                if dg is not None:
                    dg["i"] = (dg["i"] + [0, 0, 0, 0])[:13]
                    dg["i"][11] = 1
                    dg["p"] = (dg["p"] + [None] * 4)[:11]
                info["dy_packed"] += 1
        # ---- 3. activations in split storage
        if env("UZ_PACK_ACT") and not self.extra_ops:
            named = {id(v.buf) for v in self.named.values() if isinstance(v, View)}
            for b in self.bufs:
                if b.alias is not None or id(b) in tabbed or id(b) in named or (b.H * b.W) % 4 or b.N * b.H * b.W <= max(4096, self._chain_limit()):
                    continue                                 # (planes of the deep-level chain stay fp32: its convolutions split on the fly)
                wr_list, rd_list, ok = [], [], True
                for tape, ops in all_ops:
                    for o in ops:
                        for j, r in enumerate(o["p"]):
                            if not touches(r, b):
                                continue
                            c = o["code"]
                            if not isinstance(r, View) or r.nb is not None:
                                ok = False
                            elif c in self._PACK_WRITERS and j == self._PACK_WRITERS[c][0]:
                                if c == "UZ_OP_BN_RELU_FWD":
                                    i = o["i"]
                                    npx = i[3] * i[4] * i[5]
                                    one_launch = 4096 < npx <= self.L.uz_bn_fwd_fused_limit(i[4], i[5])      # mid path: a-priori bound from the parameters
                                    if not (i[6] and i[9] == 0 and (i[8] > 0 or one_launch)):
                                        ok = False           # needs the bound before the first word is written
                                if c != "UZ_OP_BN_RELU_FWD" and o["p"][2] is None:
                                    ok = False               # pooling / interpolation forward their INPUT's bound
                                wr_list.append((o, r))
                            elif c in self._PACK_READERS and j == self._PACK_READERS[c]:
                                i = o["i"]
                                kind = 0 if c == "UZ_OP_CONV_FWD" else 2
                                if i[7] != 3 or self.L.uz_conv_route(kind, i[0], i[2], i[4], i[5], i[6], 3) != 1 or (kind == 0 and i[9]):
                                    ok = False
                                rd_list.append((o, r))
                            else:
                                ok = False
                if not ok or not wr_list or not rd_list:
                    continue
                wr_list.sort(key=lambda t: t[1].c0)
                segs = [(r.c0, r.c0 + r.C) for _, r in wr_list]
                if any(a[1] > b2[0] for a, b2 in zip(segs, segs[1:])):
                    continue                                 # overlapping producers
                plan_rd = []
                for o, r in rd_list:
                    cov = [sg for sg in segs if sg[0] < r.c0 + r.C and r.c0 < sg[1]]
                    tiled = cov and cov[0][0] <= r.c0 and cov[-1][1] >= r.c0 + r.C and all(a[1] == b2[0] for a, b2 in zip(cov, cov[1:]))
                    if not tiled or len(cov) > 2 or (len(cov) == 2 and (cov[1][0] - r.c0) % 16):
                        ok = False
                        break
                    plan_rd.append((o, r, cov))
                if not ok:
                    continue
                slot_of = {}
                for o, r in wr_list:
                    vi, bi = self._PACK_WRITERS[o["code"]]
                    slot = slot_of[(r.c0, r.c0 + r.C)] = ("amax", self._new_amax())
                    o["p"][bi] = slot + ("acc",)
                    if o["code"] == "UZ_OP_BN_RELU_FWD":
                        o["i"] = (o["i"] + [0])[:11]
                        o["i"][10] = 1
                    elif o["code"] == "UZ_OP_BILINEAR_FWD":
                        o["i"] = (o["i"] + [0])[:8]
                        o["i"][7] = 1
                    else:
                        o["i"] = (o["i"] + [0])[:7]
                        o["i"][6] = 1
                for o, r, cov in plan_rd:
                    s1 = slot_of[cov[0]]
                    s2 = slot_of[cov[1]] if len(cov) == 2 else None
                    seg = cov[1][0] - r.c0 if len(cov) == 2 else 0
                    if o["code"] == "UZ_OP_CONV_FWD":
                        o["p"] = (o["p"] + [None])[:11]
                        o["p"][5], o["p"][10] = s1, s2
                        o["i"] = (o["i"] + [0, 0])[:12]
                        o["i"][10], o["i"][11] = 1, seg
                    else:
                        o["p"][5], o["p"][7] = s1, s2
                        o["i"][8], o["i"][9] = 1, seg
                b.packed = True
                info["act_packed"] += 1
                info["act_views"] += len(wr_list)
        return info

    def _bn_offchain_pass(self):
        """Takes BatchNorm's apply pass out of the forward's dependency chain (round 6; UZ_BN_OFFCHAIN=1, off by default).  For a Conv -> BatchNorm -> ReLU
        unit on the large planes whose activation is kept in split storage and read, in the forward tape, by exactly ONE 3 x 3 split-path
        convolution (the middle layers of every three-convolution block; the weight gradient of that convolution reads it too, in the
        backward tape): the unit's BatchNorm op becomes two - the statistics launch, which stays in the unit's group, and the apply pass, a
        group of its own - and the consumer reads the unit's PRE-normalisation output with its statistics table and applies BatchNorm +
        ReLU in its staging (uz_conv_fwd_bn_ex: the same operand pieces the apply pass stores).  The consumer then depends on the
        statistics launch only; the apply pass still writes the activation for the weight gradient, beside the chain instead of in it."""
        info = dict(units=0)
        self.bn_offchain = info
        if os.environ.get("UZ_BN_OFFCHAIN", "0") != "1" or self.L.uz_get_conv_math() in (0, 3) or not self.bn_training or self.extra_ops:
            return info
        ops = self.fwd_ops
        for u in [u for u in self.__dict__.get("_units", []) if "bn_fwd" in u]:
            B = u["bn_fwd"]
            i = B["i"]
            if not (len(i) > 10 and i[10] == 1 and i[8] > 0 and i[9] == 0 and i[6] == 1) or (len(i) > 11 and i[11]) or (len(i) > 13 and i[13]):
                continue                                             # large path, statistics from the convolution's partials, split-storage output
            if i[3] * i[4] * i[5] <= self.L.uz_bn_fwd_fused_limit(i[4], i[5]) or not any(o is B for o in ops):
                continue
            y, a = B["p"][0], B["p"][6]
            if not (isinstance(a, View) and isinstance(y, View)) or a.nb is not None or y.nb is not None or not a.contiguous:
                continue
            readers = [(o, j) for o in ops for j, r in enumerate(o["p"]) if o is not B and isinstance(r, View) and r.buf is a.buf]
            if len(readers) != 1 or any(isinstance(r, View) and r.buf is a.buf for o in self.loss_ops for r in o["p"]):
                continue
            R, j = readers[0]
            ri, rv = R["i"], R["p"][0]
            if R["code"] != "UZ_OP_CONV_FWD" or j != 0 or ri[7] != 3 or ri[8] or not (len(ri) > 10 and ri[10] == 1) or (len(ri) > 11 and ri[11]) or \
                    (len(ri) > 9 and ri[9]) or (len(ri) > 12 and ri[12]) or (len(ri) > 13 and ri[13]) or (rv.c0, rv.C, rv.nb) != (0, a.buf.C, None):
                continue
            k = next(n for n, o in enumerate(ops) if o is B)
            fin = dict(B, p=list(B["p"]), i=(list(i) + [0, 0])[:12], f=list(B["f"]))
            fin["p"][0] = fin["p"][6] = None
            fin["i"][11] = 1
            self._detached = getattr(self, "_detached", 0) + 1
            app = dict(B, p=list(B["p"]), i=(list(i) + [0, 0])[:12], f=list(B["f"]), gid=("bnapply", self._detached))
            app["p"][3] = app["p"][4] = app["p"][9] = None
            app["p"][8] = tuple(app["p"][8][:2])                       # reads the bound the statistics launch published
            app["i"][11] = 2
            ops[k:k + 1] = [fin, app]
            u["bn_fwd"] = fin
            R["p"] = (list(R["p"]) + [None] * 12)[:12]
            R["p"][0], R["p"][10], R["p"][11] = y, None, B["p"][5]
            R["i"] = (list(ri) + [0] * 13)[:13]
            R["i"][1], R["i"][10], R["i"][12] = y.Ctot, 0, 1 | (int(i[7]) << 1)
            info["units"] += 1
        return info

    # ------------------------------------------------------------------ bf16 storage (BASELINE config 5: PHiSeg3D "bf16")
    # tensor operand slots of the ops that have a bf16-storage form (include/uz_api.h, "bf16 storage"), in the order of the bits of i[13]
    _B16_SLOTS = {
        "UZ_OP_CONV_FWD": (0, 3), "UZ_OP_CONV_BWD_DATA": (0, 2), "UZ_OP_CONV_BWD_WEIGHT": (0, 1),
        "UZ_OP_BN_RELU_FWD": (0, 6), "UZ_OP_BN_RELU_BWD": (0, 1, 5),
        "UZ_OP_AVGPOOL3D_FWD": (0, 1), "UZ_OP_AVGPOOL3D_BWD": (0, 1), "UZ_OP_DEPTH_LERP_FWD": (0, 1), "UZ_OP_DEPTH_LERP_BWD": (0, 1),
        "UZ_OP_BILINEAR_FWD": (0, 1), "UZ_OP_BILINEAR_BWD": (0, 1),          # (only the high-resolution side: see _b16_slots)
    }

    def _b16_slots(self, o):
        """(p[] slot, bit of i[13]) of the tensor operands of op `o` that may be bf16; () = no bf16-storage form."""
        c = o["code"]
        slots = self._B16_SLOTS.get(c)
        if not slots:
            return ()
        if c.startswith("UZ_OP_CONV_") and o["i"][7] == 1:       # 1x1 head: only the many-channel side (x / dx), the 1 .. 8-channel side stays fp32
            return {"UZ_OP_CONV_FWD": ((0, 0),), "UZ_OP_CONV_BWD_DATA": ((2, 1),), "UZ_OP_CONV_BWD_WEIGHT": ((0, 0),)}[c]
        if c == "UZ_OP_BILINEAR_FWD":
            return ((1, 1),)
        if c == "UZ_OP_BILINEAR_BWD":
            return ((0, 0),)
        return tuple((j, k) for k, j in enumerate(slots))

    def _b16_ok(self, o):
        """Can op `o` run through its bf16-storage entry point (whatever the formats of its operands turn out to be)?"""
        c, i, p = o["code"], o["i"][:13], o["p"]                # (i[13] = the format bits this pass sets)
        if c in ("UZ_OP_CONV_FWD", "UZ_OP_CONV_BWD_DATA", "UZ_OP_CONV_BWD_WEIGHT") and i[7] == 1:
            kind = {"UZ_OP_CONV_FWD": 0, "UZ_OP_CONV_BWD_DATA": 1, "UZ_OP_CONV_BWD_WEIGHT": 2}[c]
            cin, cout = (i[2], i[0]) if kind == 1 else (i[0], i[2])
            if cout not in (1, 2, 3, 4, 6, 8) or cin > 512 or (i[5] * i[6]) % 4:
                return False
            if kind == 0:
                return not any(i[8:])                          # (no ReLU, no slabs)
            if kind == 1:
                return not any(i[9:]) and len(p) <= 7
            return not any(i[8:])
        if c in ("UZ_OP_CONV_FWD", "UZ_OP_CONV_BWD_DATA", "UZ_OP_CONV_BWD_WEIGHT"):
            kind = {"UZ_OP_CONV_FWD": 0, "UZ_OP_CONV_BWD_DATA": 1, "UZ_OP_CONV_BWD_WEIGHT": 2}[c]
            W = i[6]
            # (a weight gradient may leave its slabs to the table launch - i[11] - in either storage: uz_conv_bwd_weight_b16 takes slabs_out)
            if i[7] != 3 or W % 32 or (any(i[8:11]) or any(i[13:]) if kind == 2 else any(i[8:]) and kind != 1):      # (weight gradient: i[11] slabs-only, i[12] the slab count the buffer was sized for)
                return False
            if kind == 1 and (any(i[9:]) or len(p) > 7):
                return False
            if kind == 2 and p[3] is not None:
                return False
            cin, cout = (i[2], i[0]) if kind == 1 else (i[0], i[2])
            if kind != 2 and self.L.uz_conv_split_parts(kind, cin, cout, i[4], i[5], i[6]) != 1:
                return False
            return self.L.uz_conv_route(kind, cin, cout, i[4], i[5], i[6], 3) == 1
        if c == "UZ_OP_BN_RELU_FWD":
            return i[3] * i[4] * i[5] > 32768 and (i[4] * i[5]) % 4 == 0 and i[6] == 1 and not any(i[9:])
        if c == "UZ_OP_BN_RELU_BWD":
            return i[4] * i[5] * i[6] > 32768 and (i[5] * i[6]) % 4 == 0 and not any(i[8:]) and len(p) <= 11
        if c in ("UZ_OP_AVGPOOL3D_FWD", "UZ_OP_AVGPOOL3D_BWD"):
            return i[5] % 4 == 0 and i[4] % 2 == 0
        if c in ("UZ_OP_DEPTH_LERP_FWD", "UZ_OP_DEPTH_LERP_BWD"):
            return (i[4] * i[5]) % 4 == 0
        if c == "UZ_OP_BILINEAR_FWD":                          # i = [C, CtotX, CtotY, N, H, W, align_corners, (packed)]: the band kernel's shapes
            return i[5] % 4 == 0 and i[5] <= 128 and i[4] >= 4 and not any(i[7:]) and (len(p) < 3 or True)
        if c == "UZ_OP_BILINEAR_BWD":                          # i = [C, CtotDy, CtotDx, N, H, W, align_corners, acc]
            return 2 * i[5] <= 128 and i[4] >= 4 and 256 % i[5] == 0 and len(p) <= 2 and not any(i[8:])
        return False

    def _b16_pass(self):
        """Which volume tensors are kept in bf16 (UZ_STORE_B16=1 / Plan.store_b16, single-piece bf16 arithmetic only): a buffer whose
        EVERY reader and writer - in every tape - is an op with a bf16-storage form on a shape that form serves (the 3x3x3
        convolutions on the matrix pipe with planes wider than 32, the large-plane BatchNorm path, AvgPool3d, the depth stage of the
        trilinear interpolation).  The decision is per buffer, every op carries one format bit per tensor operand (i[13]), so the
        two formats mix freely: what some other op touches (the 1x1x1 heads' inputs, the in-plane interpolation's operands, the
        latent and loss tensors, the image) stays fp32.  A unit's dy moves to a bf16 twin of the zero-bordered scratch class when
        its three backward ops all qualify.  Returns / keeps a summary in self.b16_info."""
        info = self.b16_info = dict(buffers=0, grads=0, dy=0, bytes_saved=0, ops=0)
        on = os.environ.get("UZ_STORE_B16", "1" if self.__dict__.get("store_b16") else "0") == "1"
        if not on or self.L.uz_get_conv_math() != 3 or not self.bn_training or self.extra_ops:
            return info
        all_ops = [o for ops in (self.fwd_ops, self.loss_ops, self.bwd_ops) for o in ops]
        tabbed = {id(q.buf) for t in self.ptr_tables for q in t if isinstance(q, View)}
        named = {id(v.buf) for v in self.named.values() if isinstance(v, View)}

        def buf_of(r):
            if isinstance(r, View):
                return r.buf
            if isinstance(r, _ScratchView):
                return r.view.buf if r.view is not None else None
            if isinstance(r, tuple) and r and r[0] == "win":
                return r[1].buf
            return None
        ok_cache = {}
        bad, seen = set(), set()
        for o in all_ops:
            slots = [j for j, _ in self._b16_slots(o)]
            if id(o) not in ok_cache:
                ok_cache[id(o)] = bool(slots) and self._b16_ok(o)
            for j, r in enumerate(o["p"]):
                b = buf_of(r)
                if b is None:
                    continue
                seen.add(id(b))
                if not (ok_cache[id(o)] and j in slots):
                    bad.add(id(b))
        for b in self.bufs:
            if not b.vol or b.alias is not None or id(b) in tabbed or id(b) in named or id(b) in bad or id(b) not in seen:
                continue
            if b.W % 32 or (b.N - 2) * b.H * b.W <= 32768:
                continue
            b.b16 = True
            info["grads" if b.name.startswith("grad:") else "buffers"] += 1
            info["bytes_saved"] += 2 * b.numel
        # dy of the units: the lane's zero-bordered scratch class (slices, elements per slice) -> its bf16 twin
        for u in self.__dict__.get("_vol_units", []):
            B, Wg, Dg, gyv = u["bn_bwd"], u["wgrad"], u["dgrad"], u["gyv"]
            if Wg is None or gyv.zkey is None or gyv.view is not None:
                continue
            if not (self._b16_ok(B) and self._b16_ok(Wg) and (Dg is None or self._b16_ok(Dg))):
                continue
            old = gyv.zkey
            new = (old[0], old[1], 16)
            self._gyz[new] = max(self._gyz.get(new, 0), (self._gyz[old] + 1) // 2)
            gyv.zkey = new
            assert B["p"][5] == ("gyvol", old)
            B["p"][5] = ("gyvol", new)
            B["b16_dy"] = Wg["b16_dy"] = True
            if Dg is not None:
                assert Dg["p"][0] == ("gywin", old)
                Dg["p"][0] = ("gywin", new)
                Dg["b16_dy"] = True
            info["dy"] += 1
        used = set()
        for o in all_ops:
            for q in o["p"]:
                if isinstance(q, _ScratchView) and q.zkey is not None:
                    used.add(q.zkey)
                elif isinstance(q, tuple) and q and q[0] in ("gyvol", "gywin"):
                    used.add(q[1])
        for k in [k for k in self._gyz if k not in used]:
            del self._gyz[k]                                   # an fp32 class every unit has left
        # format bits
        for o in all_ops:
            slots = self._b16_slots(o)
            if not slots or not ok_cache.get(id(o)):
                continue
            bits = 0
            for j, k in slots:
                r = o["p"][j]
                b = buf_of(r)
                is16 = b is not None and b.b16
                if o.get("b16_dy") and ((o["code"] == "UZ_OP_BN_RELU_BWD" and j == 5) or (o["code"] == "UZ_OP_CONV_BWD_WEIGHT" and j == 1) or (o["code"] == "UZ_OP_CONV_BWD_DATA" and j == 0)):
                    is16 = True
                bits |= int(is16) << k
            if bits:
                o["i"] = (o["i"] + [0] * 14)[:14]
                o["i"][13] = bits
                info["ops"] += 1
        return info

    # ------------------------------------------------------------------ finalisation

    # ------------------------------------------------------------------ deep-level chain (round 6; csrc/chain.hip)
    def _chain_limit(self):
        """Largest N*H*W whose ops run inside the persistent chain launch (0: off).  Only with two-piece operands (the chain's
        convolutions are split-fp16: UZ_CONV_MATH=f32 keeps every convolution on the fp32 matrix pipe, bf16 is the volume path) and
        training-mode BatchNorm."""
        px = int(os.environ.get("UZ_CHAIN", str(self.chain_px)))
        if px <= 0 or not self.bn_training or self.L.uz_get_conv_math() in (0, 3) or self.extra_ops:
            return 0
        return px

    def _hazard_deps(self, ops):
        """Per op of `ops` (program order) the set of earlier ops it must follow (RAW, WAR, WAW over the same resources the lane
        scheduler uses; bound-slot accumulators commute with each other)."""
        hist, deps = {}, []
        for k, o in enumerate(ops):
            reads, writes, accs = self._access(o)
            d = set()
            for space, lo, hi in reads:
                for e in hist.get(space, ()):
                    if e[0] < hi and lo < e[1]:
                        if e[2] is not None:
                            d.add(e[2])
                        d.update(e[4])
            for space, lo, hi in accs:
                for e in hist.get(space, ()):
                    if e[0] < hi and lo < e[1]:
                        if e[2] is not None:
                            d.add(e[2])
                        d.update(e[3])
            for space, lo, hi in writes:
                for e in hist.get(space, ()):
                    if e[0] < hi and lo < e[1]:
                        if e[2] is not None:
                            d.add(e[2])
                        d.update(e[3])
                        d.update(e[4])
            d.discard(k)
            deps.append(d)
            for space, lo, hi in reads:
                for e in hist.setdefault(space, []):
                    if e[0] < hi and lo < e[1]:
                        e[3].add(k)
                hist[space].append([lo, hi, None, {k}, set()])
            for space, lo, hi in accs:
                for e in hist.setdefault(space, []):
                    if e[0] < hi and lo < e[1]:
                        e[4].add(k)
                hist[space].append([lo, hi, None, set(), {k}])
            for space, lo, hi in writes:
                lst = hist.setdefault(space, [])
                keep = [e for e in lst if not (lo <= e[0] and e[1] <= hi)]
                keep.append([lo, hi, k, set(), set()])
                hist[space] = keep
        return deps

    @staticmethod
    def _chain_net(o):
        """The sub-network an op belongs to (first component of the name of the tensor / parameter it is about): backward chains are
        built per sub-network, so that the prior's deep levels - which only wait for the KL gradients - run beside the likelihood's
        device-filling kernels instead of behind them."""
        c, p = o["code"], o["p"]
        ref = {"UZ_OP_BN_RELU_BWD": 1, "UZ_OP_BN_RELU_FWD": 0, "UZ_OP_AVGPOOL_BWD": 0, "UZ_OP_BILINEAR_BWD": 0,
               "UZ_OP_LATENT_BWD": 5, "UZ_OP_LATENT_HEADS_BWD_DATA": 0}.get(c)
        if c in ("UZ_OP_CONV_BWD_DATA", "UZ_OP_CONV_FWD"):
            return p[1][1].split(".")[0] if isinstance(p[1], tuple) else None
        if ref is None or not isinstance(p[ref], View):
            return None
        nm = p[ref].buf.name
        return (nm[5:] if nm.startswith("grad:") else nm).split(".")[0]

    def _chain_eligible_fwd(self, o, px):
        c, i, p = o["code"], o["i"], o["p"]
        b16 = len(i) > 13 and i[13]
        if b16:
            return False
        if c == "UZ_OP_CONV_FWD":
            cin, cout, N, H, W, ks = i[0], i[2], i[4], i[5], i[6], i[7]
            if ks != 3 or N * H * W > px or i[8] or (len(i) > 10 and i[10]) or (len(p) > 9 and p[9] is not None) or not isinstance(p[0], View) or p[0].nb is not None:
                return False
            if cin <= 4:
                return True
            return cin % 16 == 0 and cout % 32 == 0 and p[5] is not None
        if c == "UZ_OP_BN_RELU_FWD":
            return i[6] == 1 and i[8] == 0 and not (len(i) > 10 and i[10]) and i[3] * i[4] * i[5] <= px and isinstance(p[0], View) and p[0].nb is None
        if c == "UZ_OP_AVGPOOL_FWD":
            return i[3] * ((i[4] + 1) // 2) * ((i[5] + 1) // 2) <= px and not (len(i) > 6 and i[6]) and p[0].nb is None
        if c == "UZ_OP_BILINEAR_FWD":
            return i[3] * 4 * i[4] * i[5] <= px and not (len(i) > 7 and i[7]) and p[0].nb is None
        if c == "UZ_OP_LATENT_HEADS_FWD":
            return i[2] == 2 and i[3] * i[4] * i[5] <= px
        return False

    def _chain_eligible_bwd(self, o, px):
        c, i, p = o["code"], o["i"], o["p"]
        if len(i) > 13 and i[13]:
            return False
        own_dy = lambda r: isinstance(r, _ScratchView) and r.view is not None and r.zkey is None
        if c == "UZ_OP_BN_RELU_BWD":
            return (i[4] * i[5] * i[6] <= px and i[8] == 0 and i[9] == 0 and i[10] == 0 and own_dy(p[5]) and isinstance(p[0], View) and p[0].nb is None
                    and isinstance(p[1], View) and not p[1].buf.packed)
        if c == "UZ_OP_CONV_BWD_DATA":
            cout, cin, N, H, W, ks = i[0], i[2], i[4], i[5], i[6], i[7]
            fold = i[10] if len(i) > 10 else 0
            if ks != 3 or N * H * W > px or not own_dy(p[0]) or fold not in (0, 3) or (len(i) > 11 and i[11]) or p[2].nb is not None:
                return False
            if cin <= 4:
                return fold == 0
            return cout % 16 == 0 and cin % 32 == 0 and p[4] is not None
        if c == "UZ_OP_AVGPOOL_BWD":
            return i[3] * i[4] * i[5] <= px and (len(p) < 3 or p[2] is None) and p[0].nb is None
        if c == "UZ_OP_BILINEAR_BWD":
            return i[3] * 4 * i[4] * i[5] <= px and (len(p) < 3 or p[2] is None) and p[0].nb is None
        if c == "UZ_OP_LATENT_BWD":
            return isinstance(p[5], View) and p[5].C == 2 and p[5].N * p[5].H * p[5].W <= px
        if c == "UZ_OP_LATENT_HEADS_BWD_DATA":
            return i[0] == 2 and i[3] * i[4] * i[5] <= px
        return False

    def _chain_pass(self):
        """Collects the small-plane ops of a tape into UZ_OP_CHAIN ops (uz_chain_run): a sub-DAG is levelled into phases of independent
        sub-ops, every 3 x 3 convolution gets a split-K factor and a slab buffer of its own, its weights a fragment-ordered image packed
        once per tape (UZ_OP_CHAIN_PACK).  Forward tape: one chain; backward tape: one chain per sub-network (_chain_net).  The tape
        keeps a valid program order: [every op the chain depends on] [chain] [the rest]; a chain's set must be convex in the dependency
        DAG (an op outside may not sit between two inside) - offenders and what hangs behind them stay per-op launches."""
        self.chain_info = {}
        px = self._chain_limit()
        if px <= 0:
            return
        only = os.environ.get("UZ_CHAIN_NETS")                # diagnostics: "fwd,prior" = the forward chain and the prior's backward chain only
        only = None if only is None else set(only.split(","))
        jobs = [("fwd", self.fwd_ops, self._chain_eligible_fwd, None)] if (only is None or "fwd" in only) else []
        if self.bwd_ops and os.environ.get("UZ_CHAIN_BWD", "1") == "1":
            nets = []
            for o in self.bwd_ops:
                if self._chain_eligible_bwd(o, px):
                    nt = self._chain_net(o)
                    if nt not in nets:
                        nets.append(nt)
            jobs += [("bwd", self.bwd_ops, self._chain_eligible_bwd, nt) for nt in nets if only is None or nt in only]
        for which, ops, elig, net in jobs:
            if ops:
                self._chain_build(which, ops, elig, net, px)
        # the chains' weight images: one buffer and one packing launch per tape, a scheduling group of its own behind the parameter bound
        for which, imgs in self._chain_imgs.items():
            if not imgs:
                continue
            off = blk = 0
            refs = []
            for wkey, e in imgs.items():
                e["off"], e["blk0"] = off, blk
                refs += [self.P(wkey), ("chainimg", which, wkey), ("raw", e["mc"]), ("raw", e["kc"]), ("raw", e["cin"]), ("raw", e["dgrad"]), ("raw", blk)]
                off += e["bytes"]
                blk += e["blocks"]
            self._chain_imgbuf[which] = self.vec("chain_weights:" + which, off // 4)
            tape = self.fwd_ops if which == "fwd" else self.bwd_ops
            pos = next(k for k, o in enumerate(tape) if o["code"] == "UZ_OP_CHAIN")
            tape.insert(pos, dict(code="UZ_OP_CHAIN_PACK", p=[self.ptr_table(refs), ("amaxw", 0), self._chain_imgbuf[which]], i=[len(imgs), blk], f=[], n=0,
                                  gid=("chainpack", which)))

    def _chain_build(self, which, ops, elig, net, px):
        fwd = which == "fwd"
        E = [k for k, o in enumerate(ops) if elig(o, px) and (net is None or self._chain_net(o) == net)]
        if len(E) < 8:
            return
        deps = self._hazard_deps(ops)
        vkey = lambda v: (id(v.buf), v.c0) if isinstance(v, View) else (id(v.view.buf), v.view.c0)
        # pairing tables
        ywriter, bn_of_y, dgrad_of_dx, bn_of_da = {}, {}, {}, {}
        for k, o in enumerate(ops):
            c = o["code"]
            if c == "UZ_OP_CONV_FWD" and isinstance(o["p"][3], View):
                ywriter[vkey(o["p"][3])] = k
            elif c == "UZ_OP_BN_RELU_FWD":
                bn_of_y[vkey(o["p"][0])] = k
            elif c == "UZ_OP_CONV_BWD_DATA" and isinstance(o["p"][2], View):
                dgrad_of_dx.setdefault(vkey(o["p"][2]), []).append(k)
            elif c == "UZ_OP_BN_RELU_BWD" and isinstance(o["p"][0], View):
                bn_of_da[vkey(o["p"][0])] = k
        Es = set(E)
        changed = True
        while changed:
            changed = False
            for k in sorted(Es):
                o = ops[k]
                c = o["code"]
                if c == "UZ_OP_BN_RELU_FWD" and o["i"][9] > 0 and ywriter.get(vkey(o["p"][0])) not in Es:
                    Es.discard(k); changed = True                 # its convolution leaves slabs for it outside the chain
                elif c == "UZ_OP_CONV_FWD" and o["i"][9] and bn_of_y.get(vkey(o["p"][3])) not in Es:
                    Es.discard(k); changed = True
                elif c == "UZ_OP_BN_RELU_BWD" and len(o["i"]) > 11 and o["i"][11] > 0:
                    dg = dgrad_of_dx.get(vkey(o["p"][0]), [])
                    if len(dg) != 1 or dg[0] not in Es:
                        Es.discard(k); changed = True             # dA arrives as slabs of a data gradient that is not in the chain
                elif c == "UZ_OP_CONV_BWD_DATA" and len(o["i"]) > 10 and o["i"][10] == 3 and bn_of_da.get(vkey(o["p"][2])) not in Es:
                    Es.discard(k); changed = True
                elif c == "UZ_OP_LATENT_BWD" and not (k + 1 < len(ops) and ops[k + 1]["code"] == "UZ_OP_LATENT_HEADS_BWD_DATA" and (k + 1) in Es):
                    Es.discard(k); changed = True
                elif c == "UZ_OP_LATENT_HEADS_BWD_DATA" and not (k >= 1 and ops[k - 1]["code"] == "UZ_OP_LATENT_BWD" and (k - 1) in Es):
                    Es.discard(k); changed = True
            # convexity: no op outside may both depend on the set and be depended on by it
            desc = set()
            for k in range(len(ops)):
                if any(d in Es or d in desc for d in deps[k]):
                    desc.add(k)
            bad = {k for k in Es if any((d not in Es) and (d in desc) for d in deps[k])}
            if bad:
                drop = set(bad)
                for k in sorted(Es):
                    if any(d in drop for d in deps[k]):
                        drop.add(k)
                Es -= drop
                changed = True
        if len(Es) < 8:
            return
        anc = set()
        for k in reversed(range(len(ops))):
            if k in Es or k in anc:
                anc.update(d for d in deps[k] if d not in Es)
        # a scheduling group (one unit's ops, same gid, back to back) shares the lane's scratch - a unit's dy, a convolution's split-K
        # slabs - which the dependency analysis does not see: a group moves in front of the chain as a whole or not at all
        runs, a0 = [], 0
        for k in range(1, len(ops) + 1):
            if k == len(ops) or ops[k]["gid"] != ops[a0]["gid"]:
                runs.append((a0, k)); a0 = k
        grew = True
        while grew:
            grew = False
            for a0, b0 in runs:
                if any(k in anc for k in range(a0, b0)):
                    for k in range(a0, b0):
                        if k not in Es and k not in anc:
                            anc.add(k); grew = True
            for k in sorted(anc, reverse=True):
                for d in deps[k]:
                    if d not in Es and d not in anc:
                        anc.add(d); grew = True
        desc = set()
        for k in range(len(ops)):
            if any(d in Es or d in desc for d in deps[k]):
                desc.add(k)
        if any(k in desc for k in anc):
            return                                            # a group straddles the chain: leave this tape / sub-network per-op
        level = {}
        for k in sorted(Es):
            level[k] = 1 + max((level[d] for d in deps[k] if d in Es), default=-1)
        imgs = self._chain_imgs.setdefault(which, {})
        G = int(os.environ.get("UZ_CHAIN_WGS" if fwd else "UZ_CHAIN_WGS_BWD", str(self.chain_wgs if fwd else self.chain_wgs_bwd)))
        sub = []

        def emit(k, code, p, i, f=(), lvl=None):
            sub.append(dict(code=code, p=list(p), i=[int(v) for v in i], f=list(f), level=level[k] if lvl is None else lvl, orig=ops[k], k=k))

        def image(wkey, mc, kc, cin, dgrad):
            if wkey not in imgs:
                imgs[wkey] = dict(mc=mc, kc=kc, cin=cin, dgrad=dgrad, bytes=self.L.uz_chain_packed_bytes(kc, mc), blocks=self.L.uz_chain_pack_blocks(kc, mc))
            return ("chainimg", which, wkey)

        extra_levels = {}            # op index -> 1 when a slab-sum phase was put behind it (shifts everything that depends on it)
        slab_of = {}
        for k in sorted(Es, key=lambda k: (level[k], k)):
            o = ops[k]
            c, i, p = o["code"], o["i"], o["p"]
            if c == "UZ_OP_CONV_FWD":
                cin, cout, N, H, W = i[0], i[2], i[4], i[5], i[6]
                if cin <= 4:
                    emit(k, "UZ_CH_CONV3_SMALL", [p[0], p[1], p[2], p[3]], [cin, i[1], cout, i[3], N, H, W])
                    continue
                bk = bn_of_y.get(vkey(p[3]))
                S = self.L.uz_chain_conv_ksplit(cin, cout, N, H, W, G) if (bk in Es) else 1
                wkey = p[1][1]
                assert p[1][0] == "param" and p[1][2] == 0, p[1]
                slab = self.vec(wkey + ":chslab", S * N * cout * H * W) if S > 1 else None
                if slab is not None:
                    slab_of[vkey(p[3])] = (slab, S)
                emit(k, "UZ_CH_CONV3", [p[0], image(wkey, cout, cin, cin, 0), p[2] if S == 1 else None, p[3] if S == 1 else None, slab, p[5], p[6]],
                     [cin, i[1], cout, i[3], N, H, W, S, 0])
                self._packs["fwd"].pop(wkey, None)          # the image replaces the per-tape LDS image of the split kernels
            elif c == "UZ_OP_BN_RELU_FWD":
                slab, S = slab_of.get(vkey(p[0]), (None, 1))
                cbias = ops[ywriter[vkey(p[0])]]["p"][2] if S > 1 else None
                emit(k, "UZ_CH_BN_FWD", [p[0], p[1], p[2], p[3], p[4], p[5], p[6], slab, p[8], cbias], [i[0], i[1], i[2], i[3], i[4] * i[5], i[7], S], [o["f"][0], o["f"][1]])
            elif c == "UZ_OP_AVGPOOL_FWD":
                emit(k, "UZ_CH_AVGPOOL_FWD", [p[0], p[1], p[2], p[3]], i[:6])
            elif c == "UZ_OP_BILINEAR_FWD":
                emit(k, "UZ_CH_BILINEAR_FWD", [p[0], p[1], p[2], p[3]], i[:7])
            elif c == "UZ_OP_LATENT_HEADS_FWD":
                emit(k, "UZ_CH_HEADS_FWD", p[:10], [i[0], i[1], i[3], i[4] * i[5], i[6]])
            elif c == "UZ_OP_BN_RELU_BWD":
                slab, S = slab_of.get(("da",) + vkey(p[0]), (None, 1))
                emit(k, "UZ_CH_BN_BWD", [p[0], p[1], p[2], p[4], p[5].view, p[6], p[7], p[8], p[10], slab, p[3]], [i[1], i[0], i[2], i[4], i[5] * i[6], i[7], S])
            elif c == "UZ_OP_CONV_BWD_DATA":
                cout, cin, N, H, W, acc = i[0], i[2], i[4], i[5], i[6], i[8]
                wkey = p[1][1]
                assert p[1][0] == "param" and p[1][2] == 0, p[1]
                if cin <= 4:
                    emit(k, "UZ_CH_CONV3_SMALL_BWD_DATA", [p[0].view, p[1], p[2]], [cin, i[3], cout, i[1], N, H, W, acc])
                    continue
                S = self.L.uz_chain_conv_ksplit(cout, cin, N, H, W, G)
                folded = len(i) > 10 and i[10] == 3              # the consumer's BatchNorm backward adds the slabs (the only reader of a dA this op alone writes)
                slab = self.vec(wkey + ":chdslab", S * N * cin * H * W) if S > 1 else None
                emit(k, "UZ_CH_CONV3", [p[0].view, image(wkey, cin, cout, cin, 1), None, p[2] if S == 1 else None, slab, p[4], p[5]],
                     [cout, i[1], cin, i[3], N, H, W, S, acc if S == 1 else 0])
                if S > 1 and folded:
                    slab_of[("da",) + vkey(p[2])] = (slab, S)
                elif S > 1:
                    # other writers / readers of dx: a reduction phase of its own right behind the convolution
                    extra_levels[k] = 1
                    emit(k, "UZ_CH_SLAB_SUM", [slab, p[2]], [S, N, cin, i[3], H * W, acc], lvl=level[k] + 0.5)
                self._packs["bwd"].pop(wkey, None)
            elif c == "UZ_OP_AVGPOOL_BWD":
                emit(k, "UZ_CH_AVGPOOL_BWD", [p[0], p[1]], i[:7])
            elif c == "UZ_OP_BILINEAR_BWD":
                emit(k, "UZ_CH_BILINEAR_BWD", [p[0], p[1]], i[:8])
            elif c == "UZ_OP_LATENT_BWD":
                h = ops[k + 1]
                hi, hp = h["i"], h["p"]
                # (head a = the sigma head: LATENT_HEADS_BWD_DATA p = g_pre, g_mu, w_sigma, w_mu, dh)
                emit(k, "UZ_CH_LATENT_HEADS_BWD", [p[0], p[1], p[2], p[3], p[4], p[5], p[6], hp[2], hp[3], hp[4]], [hi[1], hi[2], hi[3], hi[4] * hi[5], i[0], hi[6]],
                     lvl=level[k + 1])
            elif c == "UZ_OP_LATENT_HEADS_BWD_DATA":
                pass                                              # fused into the sub-op of the LATENT_BWD in front of it
            else:
                raise AssertionError(c)
        # a slab-sum phase sits at level + 0.5: renumber the levels densely, keeping the order
        if extra_levels:
            # everything that depends (inside the set) on an op with a slab-sum must come after the half level: recompute levels with that op one deeper
            lv2 = {}
            for k in sorted(Es):
                lv2[k] = max((lv2[d] + 1 + extra_levels.get(d, 0) for d in deps[k] if d in Es), default=0)
            for e in sub:
                e["level"] = lv2[e["k"]] + (1 if e["code"] == "UZ_CH_SLAB_SUM" else 0)
                if e["code"] == "UZ_CH_LATENT_HEADS_BWD":
                    e["level"] = lv2[e["k"] + 1]
        sub.sort(key=lambda e: (e["level"], e["k"], e["code"] == "UZ_CH_SLAB_SUM"))
        dense = {lv: n for n, lv in enumerate(sorted({e["level"] for e in sub}))}
        for e in sub:
            e["level"] = dense[e["level"]]
        idx = len(self._chains)
        self._chains.append(dict(sub=sub, which=which, n_wgs=G, net=net))
        # access of the whole launch = the union of what its sub-ops touch (the packed images instead of the split kernels' ones)
        acc_ops = []
        for k in sorted(Es):
            q = dict(ops[k])
            q["p"] = list(q["p"])
            if q["code"] == "UZ_OP_CONV_FWD" and len(q["p"]) > 8:
                q["p"][8] = None
            if q["code"] == "UZ_OP_CONV_BWD_DATA" and len(q["p"]) > 6:
                q["p"][6] = None
            acc_ops.append(q)
        n_ph = 1 + max(e["level"] for e in sub)
        flops = sum(2.0 * e["i"][4] * e["i"][5] * e["i"][6] * e["i"][0] * e["i"][2] * 9 for e in sub if e["code"] == "UZ_CH_CONV3")
        chain_op = dict(code="UZ_OP_CHAIN", p=[("chaintab", idx, "ops"), ("chaintab", idx, "phases"), ("chaintab", idx, "state")],
                        i=[n_ph, G, len(sub)], f=[], n=0, gid=("chain", idx), _acc_ops=acc_ops, _which=which,
                        _cost_s=n_ph * 12e-6 + flops / 100e12)        # (first guess for the lane scheduler; tune_schedule measures it)
        new_ops = [ops[k] for k in range(len(ops)) if k in anc] + [chain_op] + [ops[k] for k in range(len(ops)) if k not in anc and k not in Es]
        ops[:] = new_ops
        self.chain_info.setdefault(which, []).append(dict(net=net, ops=len(sub), phases=n_ph, convs=sum(e["code"] == "UZ_CH_CONV3" for e in sub), workgroups=G))

    def _chain_tables(self, idx):
        """Device tables of chain `idx` (built once, when the arena exists): the sub-ops in phase order with their tile ranges, the
        phase table, the barrier state."""
        ch = self._chains[idx]
        if "dev" in ch:
            return ch["dev"]
        codes = _ffi.chain_codes()
        sub = ch["sub"]
        arr = (_ffi.uz_chain_op * len(sub))()
        phases, tile0, cur = [], 0, None
        for k, e in enumerate(sub):
            a = arr[k]
            a.code = codes[e["code"]]
            assert len(e["i"]) <= 16 and len(e["p"]) <= 12
            for j, v in enumerate(e["i"]):
                a.i[j] = int(v)
            for j, v in enumerate(e["f"]):
                a.f[j] = float(v)
            for j, r in enumerate(e["p"]):
                a.p[j] = self._resolve(r, 0)
            nt = self.L.uz_chain_op_tiles(C.byref(a))
            if nt <= 0:
                raise RuntimeError(f"chain sub-op {e['code']} {e['i']} is not covered by uz_chain_run")
            if e["level"] != cur:
                cur, tile0 = e["level"], 0
                phases.append([k, 0])
            # the op's tiles are dealt to the workgroups round-robin from workgroup tile0 on: behind a short op the next one starts
            # on the workgroups the first left idle
            a.tile0, a.ntiles = tile0 % ch["n_wgs"], nt
            tile0 += nt
            phases[-1][1] += 1
            e["tile0"], e["ntiles"] = a.tile0, nt
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        dev = dict(ops=raw.to(self.device), phases=torch.tensor(phases, dtype=torch.int32).reshape(-1).to(self.device),
                   state=torch.zeros(self.L.uz_chain_state_bytes() // 4, dtype=torch.int32, device=self.device), n_phases=len(phases))
        ch["dev"] = dev
        return dev

    def chain_status(self, stream_ptr):
        """0 when every chain launch of the last replay completed, else 1 + the phase whose grid barrier timed out (synchronises)."""
        worst = 0
        for idx, ch in enumerate(self._chains):
            if "dev" in ch:
                out = C.c_int(0)
                _ffi.check(self.L.uz_chain_status(C.c_void_p(ch["dev"]["state"].data_ptr()), C.byref(out), C.c_void_p(stream_ptr)), "chain_status")
                worst = max(worst, out.value)
        return worst

    def finalize(self, want_backward=True):
        assert not self.finalized
        self.loss_scale = self.vec("loss_scale", 1)
        if want_backward:
            for fn in self._bwd_head:
                self._newgroup()
                fn()
            for fn in reversed(self._bwd):
                self._newgroup()
                fn()
            if self._dbias_jobs:
                # all bias gradients of the folded ReLU backward in one launch (21 launches of ~7 us in U-Net's single chain)
                for part_jobs in self._by_bucket(self._dbias_jobs, lambda j: j[1]):
                    refs = [q for part, bkey, rows, c, dbl in part_jobs for q in (part, self.G(bkey), ("raw", rows), ("raw", c), ("raw", dbl))]
                    self._newgroup()
                    self._emit(self.bwd_ops, "UZ_OP_CHAN_SUM_TABLE",
                               p=[self.ptr_table(refs), ("gflat_keys", tuple(j[1] for j in part_jobs))],
                               i=[len(part_jobs), max(j[3] for j in part_jobs)])
                self._dbias_jobs = []
            jobs = self.__dict__.get("_wgrad_jobs", [])
            if jobs:
                # One launch for ALL layers.  (UZ_WGRAD_TABLE_CHUNK = n: one launch per n layers in backward order, each a scheduling group
                # that is ready as soon as ITS layers' slabs are written - measured: 1 789 / 1 790 / 1 793 / 1 791 images/s for one table /
                # chunks of 16 / 8 / 32 layers, three alternations; with the chunks PRIORITISED in the lane scheduler - started the moment their layers are done - 1 820 -> 1 795 / 1 782 for chunks
                # of 16 / 8: they delay the critical chains.  The single launch behind the tape stays.)
                chunk = int(os.environ.get("UZ_WGRAD_TABLE_CHUNK", "0"))
                chunk = len(jobs) if chunk <= 0 else chunk
                parts = self._by_bucket(jobs, lambda j: j[1]) if self.grad_buckets else [jobs[j0:j0 + chunk] for j0 in range(0, len(jobs), chunk)]
                for part in parts:
                    refs, blk = [], 0
                    for slabbuf, wkey, nslab, co, ci, kk, *vol in part:          # (vol: the C of a depth window's row, else absent)
                        refs += [slabbuf, self.G(wkey), ("raw", nslab), ("raw", co), ("raw", ci), ("raw", kk), ("raw", blk), ("raw", vol[0] if vol else 0)]
                        blk += self.L.uz_wgrad_reduce_blocks(ci, co, 3 if kk == 9 else 1)
                    self._newgroup()
                    self._emit(self.bwd_ops, "UZ_OP_WGRAD_REDUCE_TABLE",
                               p=[self.ptr_table(refs), ("gflat_keys", tuple(j[1] for j in part))], i=[len(part), blk])
                self._wgrad_jobs = []
            # the regulariser's gradients ADD to what the layers wrote (g += coeff w / |w|): behind the deferred table reductions, which
            # ASSIGN the bias / weight gradients they sum (round 4 until its last hours ran them in front - the table overwrote the
            # regulariser's share of every 3 x 3 weight gradient; tests/test_unet_probunet_gpu.py now compares the plans with and
            # without tables bit for bit)
            for fn in self._bwd_tail:
                self._newgroup()
                fn()
            # data parallel: one event per gradient bucket, recorded as soon as every writer of that slice of the flat
            # gradient buffer is done (the scheduler hoists the marker to that point of the DAG); the communication
            # stream waits for it and all-reduces the bucket beside the rest of the backward tape
            for b, (lo, hi) in enumerate(self.grad_buckets):
                self._newgroup()
                self._emit(self.bwd_ops, "UZ_OP_EVENT_RECORD", p=[("event", b), ("gflat_range", lo, hi)])
        self._bwd = []
        self._round4_passes()
        self._bn_offchain_pass()
        self._b16_pass()
        self._chain_pass()
        # magnitude-bound slots: zero the forward-side slots and measure the parameter bound at the head of the forward tape,
        # zero the backward-side slots at the head of the backward tape (group 0 of each tape: everything else depends on it)
        self.n_amax_fwd = self.n_amax
        bwd_slots = sorted(self._amax_bwd)
        fwd_slots = [k for k in range(self.n_amax) if k not in set(bwd_slots)]
        # renumber so that forward slots are [0, nf) and backward slots [nf, n): one memset each
        remap = {k: i for i, k in enumerate(fwd_slots + bwd_slots)}
        self._amax_remap, self.n_amax_fwd = remap, len(fwd_slots)
        head = self._gid = -1
        pack_ops = {}
        for which, lst in self._packs.items():
            if not lst:
                continue
            off = rows = 0
            refs = []
            for wkey, e in lst.items():
                e["off"], e["row0"] = off, rows
                off += e["bytes"]
                rows += e["rows"]
                refs += [self.P(wkey), ("packw", which, wkey), ("raw", e["mc"]), ("raw", e["kc"]), ("raw", e["cin"]), ("raw", e["cot"]),
                         ("raw", e["dgrad"] | (e["vol"] << 1)), ("raw", e["row0"])]
            self._packbuf[which] = self.vec("packed_weights:" + which, off // 4)
            # a scheduling group of its own (gid head - 1) behind the bound measurement: only the layers that READ the packed images wait for
            # it - the first layers of both encoders (1 / 3 input channels: fp32 matrix path) start beside it (40 us off the forward's chain)
            own = os.environ.get("UZ_PACK_OWN_GROUP", "1") == "1"
            pack_ops[which] = dict(code="UZ_OP_PACK_WEIGHTS", p=[self.ptr_table(refs), ("amaxw", 0), self._packbuf[which]], i=[len(lst), rows], f=[], n=0,
                                   gid=head - 1 if own else head)
        if "bwd" in pack_ops and self.bwd_ops:
            self.bwd_ops[:0] = [pack_ops["bwd"]]
        if self.fwd_ops:
            ops0 = []
            ops0.append(dict(code="UZ_OP_MEMSET", p=[("amaxrange", 0, self.n_amax_fwd)], i=[], f=[], n=4 * _AMAX_FLOATS * self.n_amax_fwd, gid=head))
            ops0.append(dict(code="UZ_OP_ABSMAX", p=[("pflat",), ("amaxw", 0)], i=[], f=[], n=self.ptab.n_params, gid=head))
            if "fwd" in pack_ops:
                ops0.append(pack_ops["fwd"])
            self.fwd_ops[:0] = ops0
        # Extra forward-only tapes (decode) run long after the forward tape that measured the parameter bound in slot 0 -
        # behind an optimiser step or a load_state_dict the bound may be stale, and a bound more than 4x too small overflows the
        # fp16 pieces.  A tape with a convolution on the split path therefore re-measures the bound at its own head.
        for name, ops in self.extra_ops.items():
            if any(o["code"] == "UZ_OP_CONV_FWD" and o["i"][7] == 3 and
                   self.L.uz_conv_route(0, o["i"][0], o["i"][2], o["i"][4], o["i"][5], o["i"][6], 3) == 1 for o in ops):
                ops[:0] = [dict(code="UZ_OP_MEMSET", p=[("amaxrange", 0, 1)], i=[], f=[], n=4 * _AMAX_FLOATS, gid=head),
                           dict(code="UZ_OP_ABSMAX", p=[("pflat",), ("amaxw", 0)], i=[], f=[], n=self.ptab.n_params, gid=head)]
        if self.bwd_ops and bwd_slots:
            self.bwd_ops[:0] = [dict(code="UZ_OP_MEMSET", p=[("amaxrange", self.n_amax_fwd, self.n_amax)], i=[], f=[],
                                     n=4 * _AMAX_FLOATS * (self.n_amax - self.n_amax_fwd), gid=head)]
        # arena layout
        off = 0
        for slot in self.__dict__.get("_rev_slots", {}).values():
            slot["off"] = off
            off += -(-slot["floats"] // _ALIGN) * _ALIGN
        for b in self.bufs:
            if b.alias is not None:
                b.off = b.alias["off"]
                continue
            b.off = off
            off += -(-b.words // _ALIGN) * _ALIGN
        # scratch regions are private to a scheduling group, so every capture lane gets its own copy
        self.gy_off, self.scratch_off, self.gyz_off = [], [], []
        for _ in range(self.n_lanes):
            self.gy_off.append(off)
            off += -(-self.scratch["gy"] // _ALIGN) * _ALIGN
            zo = {}
            for key, fl in self._gyz.items():
                zo[key] = off
                off += -(-fl // _ALIGN) * _ALIGN
            self.gyz_off.append(zo)
            so = {}
            for k in ("bn", "wgrad", "ce"):
                so[k] = off
                off += -(-(self.scratch[k] // 4 + 1) // _ALIGN) * _ALIGN
            self.scratch_off.append(so)
        if self.grad_buckets and self.device.type == "cuda":
            for _ in self.grad_buckets:
                h = C.c_void_p()
                _ffi.check(self.L.uz_event_create(C.byref(h), 0), "event_create")
                self.events.append(h.value)
        self.amax_off = off
        off += self.n_amax * _AMAX_FLOATS
        self.arena_floats = off
        self.arena = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.base = self.arena.data_ptr()
        # pointer tables
        ntab = sum(len(t) for t in self.ptr_tables)
        self.ptrtab = torch.zeros(max(ntab, 1), dtype=torch.int64, device=self.device)
        vals, self._tab_off, k = [], [], 0
        for t in self.ptr_tables:
            self._tab_off.append(k)
            for r in t:
                vals.append(self._resolve(r))
                k += 1
        if vals:
            self.ptrtab.copy_(torch.tensor(vals, dtype=torch.int64))
        self.tapes, self.scheds, self._program_order = {}, {}, {}
        for nm, ops in (("fwd", self.fwd_ops), ("loss", self.loss_ops), ("bwd", self.bwd_ops), *self.extra_ops.items()):
            self._program_order[nm] = list(ops)              # the scheduler reorders `ops` in place; reschedule() starts from this order again
            self.scheds[nm] = self._schedule(ops)
            self.tapes[nm] = self._materialize(ops)
        self.loss_scale_t = self.tensor(self.loss_scale).view(1)
        self.loss_scale_t.fill_(1.0)
        self.finalized = True
        return self

    def reschedule(self, which):
        """Schedules tape `which` again from its program order - after the per-op durations (`o["cost_us"]`, Engine.tune_schedule)
        or the scheduler's settings changed - and rebuilds its op array.  Any schedule of the same DAG gives the same bits: ordered
        pairs stay ordered, and the only unordered writers of one location are the commuting magnitude-bound maxima."""
        ops = {"fwd": self.fwd_ops, "bwd": self.bwd_ops, "loss": self.loss_ops}.get(which)
        if ops is None:
            ops = self.extra_ops[which]
        ops[:] = self._program_order[which]
        self.scheds[which] = self._schedule(ops)
        self.tapes[which] = self._materialize(ops)

    def _resolve(self, r, lane=0):
        if r is None:
            return 0
        if isinstance(r, _ScratchView):
            if r.view is not None:
                return self._resolve(r.view, lane)
            if r.zkey is not None:
                return self.base + 4 * self.gyz_off[lane][r.zkey] + (2 if len(r.zkey) > 2 else 4) * r.off
            return self.base + 4 * (self.gy_off[lane] + r.off)
        if isinstance(r, View):
            assert r.buf.off is not None
            return self.base + 4 * r.buf.off + r.buf.esz * ((r.b0 * r.buf.C + r.c0) * r.buf.H * r.buf.W)
        kind = r[0]
        if kind == "param":
            return self.ptab.pflat.data_ptr() + 4 * (self.ptab.poff[r[1]] + r[2])
        if kind == "pgrad":
            return self.ptab.gflat.data_ptr() + 4 * (self.ptab.poff[r[1]] + r[2])
        if kind == "buffer":
            return self.ptab.bflat.data_ptr() + 4 * self.ptab.boff[r[1]]
        if kind == "pflat":
            return self.ptab.pflat.data_ptr()
        if kind in ("gflat", "gflat_keys"):
            return self.ptab.gflat.data_ptr()
        if kind == "scratch":
            return self.base + 4 * self.scratch_off[lane][r[1]]
        if kind == "gyview":
            return self.base + 4 * self.gy_off[lane]
        if kind == "ptrtab":
            return self.ptrtab.data_ptr() + 8 * self._tab_off[r[1]]
        if kind == "packw":
            return self.base + 4 * self._packbuf[r[1]].buf.off + self._packs[r[1]][r[2]]["off"]
        if kind == "raw":
            return int(r[1])
        if kind == "chaintab":
            return self._chain_tables(r[1])[r[2]].data_ptr()
        if kind == "chainimg":
            return self.base + 4 * self._chain_imgbuf[r[1]].buf.off + self._chain_imgs[r[1]][r[2]]["off"]
        if kind == "win":                                    # depth window of a volume: starts one slice before the view
            v = r[1]
            return self.base + 4 * v.buf.off + v.buf.esz * (((v.b0 - 1) * v.buf.C + v.c0) * v.buf.H * v.buf.W)
        if kind == "gywin":                                  # depth window of a volume unit's dy scratch (dy lives one slice into it)
            return self.base + 4 * self.gyz_off[lane][r[1]]
        if kind == "gyvol":                                  # dy of a volume unit: the real slices start one slice in
            return self.base + 4 * self.gyz_off[lane][r[1]] + (2 if len(r[1]) > 2 else 4) * r[1][1]
        if kind == "amax":
            return self.base + 4 * (self.amax_off + _AMAX_FLOATS * self._amax_remap[r[1]])
        if kind == "amaxw":
            return self.base + 4 * (self.amax_off + _AMAX_FLOATS * self._amax_remap[0])
        if kind == "amaxrange":
            return self.base + 4 * (self.amax_off + _AMAX_FLOATS * r[1])
        if kind == "event":
            return self.events[r[1]] if self.events else 0
        if kind == "gflat_range":
            return 0
        raise ValueError(r)

    def _materialize(self, ops):
        arr = (_ffi.uz_op * max(len(ops), 1))()
        for k, o in enumerate(ops):
            e = arr[k]
            e.code = self.codes[o["code"]]
            assert len(o["i"]) <= 15 and len(o["f"]) <= 4 and len(o["p"]) <= 12, o["code"]
            for j, v in enumerate(o["i"]):
                e.i[j] = v
            for j, v in enumerate(o["f"]):
                e.f[j] = v
            e.n = o["n"]
            for j, r in enumerate(o["p"]):
                e.p[j] = self._resolve(r, o.get("lane", 0))
        return arr, len(ops)

    # ------------------------------------------------------------------ lane scheduling
    # Which p[] slots an op writes (every other slot is read).  Scratch slots are private to a
    # group and parameters are read-only inside a tape, so neither creates a dependency.
    _WRITES = {
        "UZ_OP_CONV_FWD": (3, 9), "UZ_OP_CONV_BWD_DATA": (2, 8, 9), "UZ_OP_CONV_BWD_WEIGHT": (2, 3, 8),      # (slabs-only data gradient: see _op_writes)
        "UZ_OP_BN_RELU_FWD": (3, 4, 5, 6), "UZ_OP_BN_RELU_BWD": (5, 6, 7, 8), "UZ_OP_RELU_BWD": (2, 3),
        "UZ_OP_AVGPOOL_FWD": (1,), "UZ_OP_AVGPOOL_BWD": (1, 3, 4), "UZ_OP_BILINEAR_FWD": (1,), "UZ_OP_BILINEAR_BWD": (1, 3, 4),
        "UZ_OP_NEAREST_FWD": (1,), "UZ_OP_NEAREST_BWD": (1,), "UZ_OP_SPATIAL_MEAN_FWD": (1,), "UZ_OP_SPATIAL_MEAN_BWD": (1,),
        "UZ_OP_POSTERIOR_INPUT": (2,), "UZ_OP_LATENT_FWD": (3, 4), "UZ_OP_LATENT_BWD": (5, 6),
        "UZ_OP_LATENT_HEADS_FWD": (6, 7, 8, 9), "UZ_OP_LATENT_HEADS_BWD_DATA": (4,), "UZ_OP_LATENT_HEADS_BWD_WEIGHT": (3, 4, 5, 6),
        "UZ_OP_KL_FWD": (4,), "UZ_OP_KL_BWD": (5, 6, 7, 8), "UZ_OP_CE_FWD": (2,), "UZ_OP_CE_BWD": (1,),
        "UZ_OP_SUM_TERMS": (1,), "UZ_OP_SCALE": (0,), "UZ_OP_COPY": (0,), "UZ_OP_MEMSET": (0,),
        "UZ_OP_L2_NORMS": (2,), "UZ_OP_L2_NORMS_BWD": (4,),
        "UZ_OP_BCAST_CHANNELS": (1,), "UZ_OP_BCAST_CHANNELS_BWD": (1,), "UZ_OP_EVENT_RECORD": (0,), "UZ_OP_ABSMAX": (1,),
        "UZ_OP_ADD_VIEWS": (2,), "UZ_OP_W3D_PERMUTE": (1,), "UZ_OP_AVGPOOL3D_FWD": (1,), "UZ_OP_AVGPOOL3D_BWD": (1,),
        "UZ_OP_DEPTH_LERP_FWD": (1,), "UZ_OP_DEPTH_LERP_BWD": (1,), "UZ_OP_NEAREST3D_FWD": (1,), "UZ_OP_NEAREST3D_BWD": (1,),
        "UZ_OP_ABSMAX_COPY": (), "UZ_OP_PACK_WEIGHTS": (2,), "UZ_OP_CHAIN_PACK": (2,), "UZ_OP_CHAN_SUM_PARTIALS": (1,), "UZ_OP_CHAN_SUM_TABLE": (1,), "UZ_OP_WGRAD_REDUCE_TABLE": (1,),
    }

    def _op_writes(self, o):
        """p[] slots op `o` writes: the static table, except for a slabs-only data gradient (i[10] == 3: partial sums into p[7], its
        gradient view p[2] is NOT written - the consumer's BatchNorm backward reads the slabs instead)."""
        if o["code"] == "UZ_OP_CONV_BWD_DATA" and len(o["i"]) > 10 and o["i"][10] == 3:
            return (7,)
        if o["code"] == "UZ_OP_BN_RELU_FWD" and len(o["i"]) > 11 and o["i"][11]:
            return (3, 4, 5) if o["i"][11] == 1 else (6,)          # phase 1: statistics table + running buffers; phase 2: the activation only
        return self._WRITES[o["code"]]

    def _access(self, o):
        """(reads, writes, accumulates) of op `o` as resource ranges.  Accumulates (bound slots only) commute with each other
        and conflict with reads and writes."""
        if o["code"] == "UZ_OP_CHAIN":
            reads, writes, accs = [], [], []
            for q in o["_acc_ops"]:
                r_, w_, a_ = self._access(q)
                reads.extend(r_); writes.extend(w_); accs.extend(a_)
            reads.extend(self._resources(("chainimg", o["_which"], None)))
            # (inside the launch the accumulators of a bound slot are ordered against its readers by the phases; towards the rest of the
            #  tape the launch both accumulates into and reads those slots: declare them written)
            writes.extend(accs)
            return reads, writes, []
        wr = self._op_writes(o)
        reads, writes, accs = [], [], []
        for j, r in enumerate(o["p"]):
            if j in wr:
                writes.extend(self._resources(r))
            elif isinstance(r, tuple) and len(r) == 3 and r[0] == "amax" and r[2] == "acc":
                accs.extend(self._resources(r))
            else:
                reads.extend(self._resources(r))
        return reads, writes, accs

    def _resources(self, r):
        """Dependency-relevant resources behind one pointer ref: (space, lo, hi) half-open ranges."""
        if isinstance(r, _ScratchView) and r.view is not None:
            return self._resources(r.view)
        if r is None or isinstance(r, _ScratchView):
            return []
        if isinstance(r, View):
            if r.buf.alias is not None:                  # pooled scratch: differently shaped Bufs overlap in memory -> whole-region hazards
                return [(("pool", id(r.buf.alias)), 0, 1 << 30)]
            return [(("buf", id(r.buf)), r.c0, r.c0 + r.C)]
        kind = r[0]
        if kind == "pgrad":
            lo = self.ptab.poff[r[1]]
            return [(("gflat",), lo, lo + _numel(self.ptab.shape[r[1]]))]
        if kind == "gflat":
            return [(("gflat",), 0, 1 << 62)]
        if kind == "gflat_keys":
            return [(("gflat",), self.ptab.poff[k], self.ptab.poff[k] + _numel(self.ptab.shape[k])) for k in r[1]]
        if kind == "buffer":
            return [(("bnbuf", r[1]), 0, 1)]
        if kind == "ptrtab":
            return [x for q in self.ptr_tables[r[1]] for x in self._resources(q)]
        if kind == "packw":
            return self._resources(self._packbuf[r[1]]) if r[1] in self._packbuf else [(("packw", r[1]), 0, 1)]      # (before finalize() lays the images out: _chain_pass)
        if kind == "chainimg":
            return self._resources(self._chain_imgbuf[r[1]]) if r[1] in self._chain_imgbuf else [(("chainimg", r[1]), 0, 1)]
        if kind == "chaintab":
            return []
        if kind == "win":
            return self._resources(r[1])
        if kind in ("gywin", "gyvol"):
            return []
        if kind == "amax":
            # Bound slots are atomic-max accumulated by the kernels that write the data they bound.  A slot has three access
            # classes (_access): the zeroing / parameter-bound measurement at the head of a tape WRITES it; producers ACCUMULATE
            # (refs tagged "acc": unordered among themselves - no false serialisation between the producers of one concat buffer);
            # consumers READ it - and a reader is ordered against EVERY accumulator of its slot, also one that writes channels the
            # reader never touches: a kernel derives its operand scale from the slot more than once (workgroup by workgroup, main
            # loop and epilogue), so the value must not move while it runs.
            return [(("amax",), r[1], r[1] + 1)]
        if kind == "amaxw":
            return [(("amax",), 0, 1)]
        if kind == "amaxrange":
            inv = {v: k for k, v in self._amax_remap.items()}
            return [(("amax",), inv[j], inv[j] + 1) for j in range(r[1], r[2])]
        if kind == "gflat_range":
            return [(("gflat",), r[1], r[2])]
        if kind == "event":
            return [(("event", r[1]), 0, 1)]
        if kind in ("param", "pflat", "scratch", "gyview", "raw"):
            return []
        raise ValueError(r)

    @staticmethod
    def _op_cost(o):
        """Rough duration (s) of one op on MI355X - only the RELATIVE sizes matter: the list scheduler
        below uses them to decide which chains can share a lane."""
        c, i = o["code"], o["i"]
        if "_cost_s" in o:
            return o["_cost_s"]
        if c in ("UZ_OP_CONV_FWD", "UZ_OP_CONV_BWD_DATA", "UZ_OP_CONV_BWD_WEIGHT"):
            cin, cout, N, H, W, ks = i[0], i[2], i[4], i[5], i[6], i[7]
            flop = 2.0 * N * H * W * cin * cout * ks * ks
            # measured fp32-equivalent TFLOP/s per plane size at batch 32 (tools/op_profile.py), forward / data
            # gradient and weight gradient; small channel counts and 1x1 kernels run far below these
            px = N * H * W
            table = ((262144, 180, 150), (65536, 190, 159), (16384, 148, 122), (4096, 97, 85), (1024, 37, 31), (256, 11, 9), (0, 3, 2.2))
            rate = next((f if c != "UZ_OP_CONV_BWD_WEIGHT" else w) for lim, f, w in table if px >= lim) * 1e12
            if min(cin, cout) < 32 or ks == 1:
                rate = min(rate, 20e12)
            return flop / rate + 4e-6
        if c == "UZ_OP_BN_RELU_FWD":
            if len(i) > 11 and i[11] == 1:
                return 6e-6                                          # statistics from the convolution's partials: one small launch
            return i[0] * i[3] * i[4] * i[5] * 12.0 / 4e12 + 10e-6
        if c in ("UZ_OP_BN_RELU_BWD", "UZ_OP_RELU_BWD"):
            return i[1] * i[4] * i[5] * i[6] * 20.0 / 4e12 + 14e-6
        if c in ("UZ_OP_AVGPOOL_FWD", "UZ_OP_AVGPOOL_BWD", "UZ_OP_BILINEAR_FWD", "UZ_OP_BILINEAR_BWD", "UZ_OP_NEAREST_FWD", "UZ_OP_NEAREST_BWD"):
            return i[0] * i[3] * i[4] * i[5] * 24.0 / 2e12 + 6e-6
        return 6e-6

    @staticmethod
    def _op_cost_beside(o):
        """Duration (s) of one op as the lane replay REALISES it beside the other lanes' kernels (round 5: fitted per op class to the
        per-op lane traces profiles/r5_lane_trace_{fwd,bwd}.json of the headline configuration, tools/sched_calibrate.py prints the fit).
        The isolated model above is 2 - 6x short for the small launches (a 7-us convolution takes 33 - 45 us beside a device-filling
        kernel) and 1.4 - 10x long for the large ones: the list scheduler then queued off-critical weight gradients in front of a
        critical chain on the same lane and the chain started 2 ms late."""
        c = o["code"]
        if "_cost_s" in o:
            return o["_cost_s"]
        m = Plan._op_cost(o) * 1e6
        if c in ("UZ_OP_CONV_FWD", "UZ_OP_CONV_BWD_DATA"):
            us = 28.0 + 0.62 * m
        elif c == "UZ_OP_CONV_BWD_WEIGHT":
            us = 25.0 + 0.70 * m
        elif c == "UZ_OP_BN_RELU_FWD":
            us = 20.0 + 0.50 * m
        elif c in ("UZ_OP_BN_RELU_BWD", "UZ_OP_RELU_BWD"):
            us = 25.0 + 0.80 * m
        elif c in ("UZ_OP_AVGPOOL_FWD", "UZ_OP_AVGPOOL_BWD"):
            us = 12.0 + 0.06 * m
        elif c in ("UZ_OP_BILINEAR_FWD", "UZ_OP_BILINEAR_BWD", "UZ_OP_NEAREST_FWD", "UZ_OP_NEAREST_BWD"):
            us = 18.0 + 0.42 * m
        elif c == "UZ_OP_WGRAD_REDUCE_TABLE":
            us = 8.0 + 0.0031 * o["i"][1]                 # HBM-bound: PHiSeg's 106 layers = 95 301 blocks take 300 us
        elif c in ("UZ_OP_MEMSET",):
            us = 8.0
        elif c in ("UZ_OP_W3D_PERMUTE", "UZ_OP_AVGPOOL3D_FWD", "UZ_OP_AVGPOOL3D_BWD", "UZ_OP_DEPTH_LERP_FWD", "UZ_OP_DEPTH_LERP_BWD",
                   "UZ_OP_NEAREST3D_FWD", "UZ_OP_NEAREST3D_BWD", "UZ_OP_ADD_VIEWS"):
            # volume ops stream whole volumes: a flat 25 us was wrong by orders of magnitude (PHiSeg3D's static schedule)
            us = 12.0 + Plan._vol_bytes(o) / 3.0e6                  # ~3 TB/s realised beside the other lanes' kernels
        else:
            us = 25.0
        return us * 1e-6

    @staticmethod
    def _vol_bytes(o):
        """Bytes a volume / view op moves (input + output once), from its i[] (C, ..., D, H, W in the layouts of include/uz_api.h)."""
        c, i = o["code"], o["i"]
        if c == "UZ_OP_W3D_PERMUTE":
            return 8.0 * i[0] * i[1] * 27
        if c == "UZ_OP_ADD_VIEWS":
            return 12.0 * i[3] * i[4] * i[5] * i[6]
        C_, D, H, W = i[0], i[3], i[4], i[5]
        n = float(C_) * D * H * W
        if c in ("UZ_OP_AVGPOOL3D_FWD", "UZ_OP_AVGPOOL3D_BWD"):
            return 4.0 * n * (1.0 + 0.125)
        if c in ("UZ_OP_DEPTH_LERP_FWD", "UZ_OP_DEPTH_LERP_BWD"):
            return 4.0 * n * 3.0
        return 4.0 * n * (1.0 + float(i[6]) * i[6] * (i[7] if len(i) > 7 else 1))

    @staticmethod
    def _op_heavy(o):
        """True for ops that fill the chip on their own (only one of those is simulated in flight at a time)."""
        c, i = o["code"], o["i"]
        if c in ("UZ_OP_CONV_FWD", "UZ_OP_CONV_BWD_DATA", "UZ_OP_CONV_BWD_WEIGHT"):
            return i[4] * i[5] * i[6] >= 2048 and i[0] * i[2] >= 32 * 32      # measured: 512..8192 pixels all within 1 %
        if c in ("UZ_OP_BN_RELU_FWD", "UZ_OP_BN_RELU_BWD", "UZ_OP_RELU_BWD"):
            C, N, H, W = (i[0], i[3], i[4], i[5]) if c == "UZ_OP_BN_RELU_FWD" else (i[1], i[4], i[5], i[6])
            return C * N * H * W >= 4.0e6
        if c in ("UZ_OP_AVGPOOL_FWD", "UZ_OP_AVGPOOL_BWD", "UZ_OP_BILINEAR_FWD", "UZ_OP_BILINEAR_BWD", "UZ_OP_NEAREST_FWD", "UZ_OP_NEAREST_BWD"):
            return i[0] * i[3] * i[4] * i[5] >= 2.0e6
        return False

    def _schedule(self, ops):
        """Turns the tape into a DAG schedule for uz_graph_create_lanes and REORDERS `ops` in place.

        1. Hazard analysis in program order: a group (one layer's forward, or one layer's backward;
           its ops stay back to back) depends on every earlier group it has a RAW, WAR or WAW conflict
           with (buffer channel ranges, gradient ranges, BN running buffers).
        2. Event-driven list scheduling in simulated time (cost model above) on n_lanes lanes (a lane is a
           scratch copy, so the groups of a lane stay ordered): among the groups that are ready, the one
           with the longest remaining critical path starts first; at most ONE device-filling ("heavy")
           group is in flight at a time, light groups start whenever a lane is free.  This staggers the
           independent chains (posterior / prior encoders, likelihood branches) so that the latency-bound
           deep levels of one run beside the device-filling layers of another, instead of both being
           deep - and the chip idle - at the same time.
        3. The tape is rewritten in simulated start order (a topological order of the DAG, so the
           eager runner stays correct) and every group gets: its lane, and the cross-lane groups it
           must wait for that are not already implied by its lane predecessor.
        Returns the ctypes uz_sched array (indexed like the reordered ops)."""
        n, K = len(ops), self.n_lanes
        sched = (_ffi.uz_sched * max(n, 1))()
        if n == 0:
            return sched
        groups = []                                         # [first, last] op index per group, program order
        for k, o in enumerate(ops):
            if groups and ops[groups[-1][1]]["gid"] == o["gid"]:
                groups[-1][1] = k
            else:
                groups.append([k, k])
        G = len(groups)
        # ---- 1. hazards
        hist = {}                                           # space -> list of [lo, hi, last_writer_gi, readers{gi}, accumulators{gi}]
        deps = []
        for gi, (a, b) in enumerate(groups):
            reads, writes, accs = [], [], []
            for o in ops[a:b + 1]:
                r_, w_, a_ = self._access(o)
                reads.extend(r_); writes.extend(w_); accs.extend(a_)
            d = set()
            for space, lo, hi in reads:                     # after the last writer and every accumulator since
                for e in hist.get(space, ()):
                    if e[0] < hi and lo < e[1]:
                        if e[2] is not None:
                            d.add(e[2])
                        d.update(e[4])
            for space, lo, hi in accs:                      # after the last writer and every reader since (not after other accumulators)
                for e in hist.get(space, ()):
                    if e[0] < hi and lo < e[1]:
                        if e[2] is not None:
                            d.add(e[2])
                        d.update(e[3])
            for space, lo, hi in writes:
                for e in hist.get(space, ()):
                    if e[0] < hi and lo < e[1]:
                        if e[2] is not None:
                            d.add(e[2])
                        d.update(e[3])
                        d.update(e[4])
            d.discard(gi)
            deps.append(d)
            for space, lo, hi in reads:
                for e in hist.setdefault(space, []):
                    if e[0] < hi and lo < e[1]:
                        e[3].add(gi)
                hist[space].append([lo, hi, None, {gi}, set()])
            for space, lo, hi in accs:
                for e in hist.setdefault(space, []):
                    if e[0] < hi and lo < e[1]:
                        e[4].add(gi)
                hist[space].append([lo, hi, None, set(), {gi}])
            for space, lo, hi in writes:
                lst = hist.setdefault(space, [])
                keep = [e for e in lst if not (lo <= e[0] and e[1] <= hi)]     # fully covered entries are superseded
                keep.append([lo, hi, gi, set(), set()])
                hist[space] = keep
        # ---- 2. list scheduling in simulated time
        # UZ_SCHED_COST: "alone" = the isolated-launch model, "beside" = the model fitted to the lane replay's realised durations
        cmodel = os.environ.get("UZ_SCHED_COST", self.__dict__.get("sched_cost", "alone"))
        assert cmodel in ("alone", "beside"), cmodel
        op_cost = self._op_cost_beside if cmodel == "beside" else self._op_cost
        cost = [sum((o["cost_us"] * 1e-6 if "cost_us" in o else op_cost(o)) for o in ops[a:b + 1]) for a, b in groups]    # cost_us: measured (tune_schedule)
        succ = [[] for _ in range(G)]
        indeg = [len(d) for d in deps]
        for gi, d in enumerate(deps):
            for e in d:
                succ[e].append(gi)
        ready_at = [0.0] * G
        ready = [gi for gi in range(G) if indeg[gi] == 0]
        free = [0.0] * K
        tail = [None] * K
        finish = [0.0] * G
        order, lane_of = [], [None] * G
        # one device-filling group at a time (critical path first); light groups start whenever a lane is free
        # UZ_SCHED_HEAVY: "r4" = the round-4 rule (any op of the group above the size thresholds of _op_heavy), a number = groups whose
        # modelled duration reaches that many microseconds, "off" = no group is heavy (plain critical-path list scheduling)
        hmode = os.environ.get("UZ_SCHED_HEAVY", self.__dict__.get("sched_heavy", "r4"))
        if hmode == "r4":
            heavy = [any(self._op_heavy(o) for o in ops[a:b + 1]) for a, b in groups]
        elif hmode == "off":
            heavy = [False] * G
        else:
            heavy = [c >= float(hmode) * 1e-6 for c in cost]
        # (round 5, measured under the lane replay and removed - commit 0cbd7b7 has the code: device-filling groups on a lane of their own
        #  -7 %, weight-gradient-only groups on a low-priority lane -11 %, heavy weight gradients deferred behind the data-gradient chain -2 %;
        #  profiles/NOTES_r5.md section 3)
        blevel = [0.0] * G
        for gi in reversed(range(G)):
            blevel[gi] = cost[gi] + max((blevel[sg] for sg in succ[gi]), default=0.0)
        for gi, (a, b) in enumerate(groups):
            if any(o["code"] == "UZ_OP_EVENT_RECORD" for o in ops[a:b + 1]):
                blevel[gi], cost[gi] = 1e9, 1e-7      # bucket-final markers cost nothing and must fire the moment their bucket is done
                # ... and so must the bucket's deferred reductions (slab / bias-row tables): they are the last writers of the bucket, they
                # have no other successor, and left at their own bottom level the list scheduler runs them LAST - all buckets final at 98 %
                # of the tape, nothing of the exchange overlapped (headline plan under the round-5 cost model; 84 % under the old one)
                for d in deps[gi]:
                    if any(o["code"] in ("UZ_OP_WGRAD_REDUCE_TABLE", "UZ_OP_CHAN_SUM_TABLE") for o in ops[groups[d][0]:groups[d][1] + 1]):
                        blevel[d] = 1e8
        t, running = 0.0, []
        while len(order) < G:
            running = [(f, g) for f, g in running if f > t + 1e-12]
            busy_lanes = {lane_of[g] for _, g in running}
            heavy_busy = any(heavy[g] for _, g in running)
            started = False
            for gi in sorted((g for g in ready if ready_at[g] <= t + 1e-12), key=lambda g: (-blevel[g], g)):
                lanes_free = [l for l in range(K) if l not in busy_lanes]
                if not lanes_free or (heavy[gi] and heavy_busy):
                    continue
                pref = [l for l in lanes_free if tail[l] is not None and tail[l] in deps[gi]]
                lane = pref[0] if pref else min(lanes_free, key=lambda l: (free[l], l))
                lane_of[gi] = lane
                finish[gi] = t + cost[gi]
                free[lane] = finish[gi]
                tail[lane] = gi
                order.append(gi)
                ready.remove(gi)
                running.append((finish[gi], gi))
                busy_lanes.add(lane)
                heavy_busy = heavy_busy or heavy[gi]
                started = True
                for sgi in succ[gi]:
                    indeg[sgi] -= 1
                    ready_at[sgi] = max(ready_at[sgi], finish[gi])
                    if indeg[sgi] == 0:
                        ready.append(sgi)
            if not started:
                nxt = [f for f, _ in running] + [ready_at[g] for g in ready if ready_at[g] > t + 1e-12]
                assert nxt, "dependency cycle in the tape"
                t = min(nxt)
        assert len(order) == G, "dependency cycle in the tape"
        # ---- 3. rewrite the tape in schedule order, compute waits
        new_ops, first, last = [], [0] * G, [0] * G
        for gi in order:
            a, b = groups[gi]
            first[gi] = len(new_ops)
            new_ops.extend(ops[a:b + 1])
            last[gi] = len(new_ops) - 1
        ops[:] = new_ops
        sched = (_ffi.uz_sched * max(n, 1))()
        anc = [0] * G                                       # bitset of transitive predecessors (incl. lane order)
        ltail = [None] * K
        for gi in order:
            lane = lane_of[gi]
            mask = 0
            for d in deps[gi]:
                mask |= anc[d] | (1 << d)
            implied = 0
            if ltail[lane] is not None:
                implied = anc[ltail[lane]] | (1 << ltail[lane])
                mask |= implied
            per_lane = {}
            for d in deps[gi]:
                if lane_of[d] != lane and not (implied >> d) & 1:
                    cur = per_lane.get(lane_of[d])
                    if cur is None or first[d] > first[cur]:
                        per_lane[lane_of[d]] = d
            waits = list(per_lane.values())
            waits = [d for d in waits if not any(d != e and (anc[e] >> d) & 1 for e in waits)]
            anc[gi] = mask
            ltail[lane] = gi
            k0 = first[gi]
            sched[k0].n_wait = len(waits)
            for w, d in enumerate(sorted(waits, key=lambda d: first[d])):
                sched[k0].wait[w] = last[d]
                sched[last[d]].signal = 1
            for k in range(first[gi], last[gi] + 1):
                sched[k].lane = lane
                ops[k]["lane"] = lane
        # the scheduled DAG, for diagnostics (tools/critical_path.py): per group in schedule order its op range, lane and predecessors
        pos = {gi: n_ for n_, gi in enumerate(order)}
        self.__dict__.setdefault("dag", {})[id(ops)] = [dict(first=first[gi], last=last[gi], lane=lane_of[gi], deps=sorted(pos[d] for d in deps[gi]),
                                                              sim_start=finish[gi] - cost[gi], sim_cost=cost[gi], blevel=blevel[gi])
                                                         for gi in order]
        return sched

    # ------------------------------------------------------------------ execution helpers
    def tensor(self, v):
        """torch view (storage only) of a plan buffer slice, NCHW (volumes: the D real slices, i.e. DCHW)."""
        b = v.buf
        if b.b16:
            full = self.arena[b.off:b.off + b.words].view(torch.bfloat16)[:b.numel].view(b.N, b.C, b.H, b.W)
        else:
            full = self.arena[b.off:b.off + b.numel].view(b.N, b.C, b.H, b.W)
        if v.nb is not None:
            full = full[v.b0:v.b0 + v.nb]
        return full[:, v.c0:v.c0 + v.C]

    def span(self, views):
        """One flat tensor covering the given whole, consecutively allocated buffers (alignment gaps included), or None when
        they are not laid out that way: lets the host fill all of them (latent noise) with ONE launch."""
        bufs = [v.buf for v in views]
        if any(v.nb is not None or not v.contiguous or b.alias is not None for v, b in zip(views, bufs)):
            return None
        for a, b in zip(bufs, bufs[1:]):
            if not (a.off + a.numel <= b.off <= a.off + a.numel + _ALIGN):
                return None
        return self.arena[bufs[0].off:bufs[-1].off + bufs[-1].numel]

    def run(self, which, stream_ptr):
        arr, n = self.tapes[which]
        if n:
            _ffi.check(self.L.uz_run_tape(arr, n, C.c_void_p(stream_ptr)), f"tape '{which}'")

    def summary(self):
        cnt = {}
        for nm, ops in (("fwd", self.fwd_ops), ("loss", self.loss_ops), ("bwd", self.bwd_ops)):
            cnt[nm] = len(ops)
        cnt["arena_MB"] = self.arena_floats * 4 / 2 ** 20 if self.finalized else None
        return cnt


class _ScratchView:
    """Shape carrier for the shared dy scratch (gradient w.r.t. a conv output, consumed at once)."""
    def __init__(self, N, C, H, W, amax=None):
        self.N, self.C, self.H, self.W, self.Ctot = N, C, H, W, C
        self.buf = None
        self.amax = amax          # magnitude-bound slot of this unit's dy
        self.off = 0              # float offset inside the scratch (volumes: one slice, behind the zeroed border slice)
        self.nb = None
        self.view = None          # a buffer of its own instead of the lane's scratch (decoupled weight gradients, see _conv_bwd)
        self.zkey = None          # volume units: the zero-bordered scratch class (slices, floats per slice) this dy lives in
