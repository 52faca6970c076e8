"""Fused Adam over the flat parameter buffer (reference harness: torch.optim.Adam(lr=1e-3,
weight_decay=1e-5), train_model.py:49,122).  One HIP launch per contiguous run of parameters that
have a gradient; parameters whose ``grad is None`` are skipped entirely - no moment update, no
weight decay - exactly like torch.optim.Adam.  It subclasses
``torch.optim.Optimizer`` so that ``ReduceLROnPlateau`` (train_model.py:50-51) can drive ``lr``."""
import ctypes as C

import torch

from . import _ffi


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        pt = model._ptab
        self.exp_avg = torch.zeros_like(pt.pflat)
        self.exp_avg_sq = torch.zeros_like(pt.pflat)
        self.step_count = 0
        self._runs_cache = {}

    def _runs(self):
        """Contiguous [lo, hi) float ranges of the flat buffer whose parameters currently have a gradient."""
        pt = self.model._ptab
        sig = tuple(p.grad is not None for p in self.model._pmap.values())
        runs = self._runs_cache.get(sig)
        if runs is None:
            runs = []
            for (key, p), has in zip(self.model._pmap.items(), sig):
                if not has:
                    continue
                lo = pt.poff[key]
                hi = lo + p.numel()
                if runs and runs[-1][1] == lo:
                    runs[-1][1] = hi
                else:
                    runs.append([lo, hi])
            self._runs_cache[sig] = runs
        return runs

    @torch.no_grad()
    def step(self, closure=None):
        pt = self.model._ptab
        g = self.param_groups[0]
        # gradients normally ARE views of the flat gradient buffer; copy in foreign ones
        for key, p in self.model._pmap.items():
            pg = p.grad
            if pg is not None:
                gv = pt.gview(key)
                if pg is not gv and pg.data_ptr() != gv.data_ptr():
                    gv.copy_(pg)
        self.step_count += 1
        L = _ffi.lib()
        st = C.c_void_p(self.model._stream())
        for lo, hi in self._runs():
            _ffi.check(L.uz_adam_step(pt.pflat.data_ptr() + 4 * lo, pt.gflat.data_ptr() + 4 * lo,
                                      self.exp_avg.data_ptr() + 4 * lo, self.exp_avg_sq.data_ptr() + 4 * lo,
                                      hi - lo, self.step_count, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                      g["weight_decay"], 1.0, st), "adam_step")
        return None

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad()
