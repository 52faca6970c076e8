"""Helpers for the -m gpu parity tests: call libuz_hip.so through its C ABI with torch tensors as
storage and compare with a plain fp32 PyTorch CPU reference of the same op."""
import ctypes as C

import numpy as np
import torch

import unet_zoo_amd  # noqa: F401
from unet_zoo_amd import _ffi


def dev():
    return torch.device("cuda", 0)


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Call a C-ABI function; tensors become device pointers, None -> NULL; stream is appended."""
    L = _ffi.lib()
    conv = []
    for a in args:
        if isinstance(a, torch.Tensor):
            conv.append(a.data_ptr())
        else:
            conv.append(a)
    fn = getattr(L, name)
    assert len(conv) + 1 == len(fn.argtypes), f"{name}: {len(conv) + 1} arguments for a {len(fn.argtypes)}-argument entry point"
    rc = fn(*conv, stream())
    _ffi.check(rc, name)
    torch.cuda.synchronize()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def view_in(t, ctot, c0):
    """Embed an NCHW CPU tensor as channels [c0, c0+C) of a wider GPU buffer filled with NaN canaries;
    returns (buffer, slice_view_pointer_tensor)."""
    n, c, h, w = t.shape
    buf = torch.full((n, ctot, h, w), float("nan"), device=dev())
    buf[:, c0:c0 + c] = t.to(dev())
    return buf, buf[:, c0:]


def relerr(got, ref):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


def maxabs(got, ref):
    return float((got.detach().cpu().double() - ref.detach().cpu().double()).abs().max())


def L():
    """The loaded C-ABI library (size queries: uz_*_workspace)."""
    return _ffi.lib()
