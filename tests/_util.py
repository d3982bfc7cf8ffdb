"""Shared helpers for the parity tests (host-side plumbing only)."""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from anatomix_amd import _lib

TORCH_T = {"f16": torch.float16, "bf16": torch.bfloat16}


def to_ndhwc(x_ncdhw, dtype):
    return x_ncdhw.permute(0, 2, 3, 4, 1).contiguous().to(dtype)


def from_ndhwc(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def max_rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def run_conv(device, x0, x1, w, scale, shift, act, precision, planar=False, slope=0.3):
    """Calls amx_conv3d_k3_reflect.  x0/x1: NCDHW float CPU tensors (x1 half resolution or None)."""
    lib = _lib.load()
    tdt = TORCH_T[precision]
    n, c0, d, h, ww = x0.shape
    cout = w.shape[0]
    c1 = 0 if x1 is None else x1.shape[1]
    dx0 = to_ndhwc(x0, tdt).to(device)
    dx1 = None if x1 is None else to_ndhwc(x1, tdt).to(device)
    dw = w.reshape(cout, c0 + c1, 27).contiguous().float().to(device)
    dsc = None if scale is None else scale.float().to(device)
    dsh = None if shift is None else shift.float().to(device)
    wpk = torch.empty(lib.amx_conv3d_packed_bytes(c0 + c1, cout), dtype=torch.uint8, device=device)
    if planar:
        out = torch.full((n, cout, d, h, ww), float("nan"), dtype=torch.float32, device=device)
        o16, o32 = None, out
    else:
        out = torch.full((n, d, h, ww, cout), float("nan"), dtype=tdt, device=device)
        o16, o32 = out, None
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _lib.check(lib.amx_conv3d_k3_reflect(_lib.ptr(dx0), c0, _lib.ptr(dx1), c1, _lib.ptr(dw), _lib.ptr(dsc),
                                         _lib.ptr(dsh), cout, n, d, h, ww, act, slope, _lib.PRECISION[precision],
                                         _lib.ptr(wpk), _lib.ptr(o16), _lib.ptr(o32), st))
    torch.cuda.synchronize(device)
    out = out.cpu()
    return out if planar else from_ndhwc(out.float())


def ref_conv(x0, x1, w, scale, shift, act, precision, slope=0.3):
    """CPU reference with the SAME input/weight rounding as the kernel, fp32 accumulation."""
    tdt = TORCH_T[precision]
    q = lambda t: t.to(tdt).float()
    x = q(x0)
    if x1 is not None:
        x = torch.cat((x, F.interpolate(q(x1), scale_factor=2, mode="nearest")), dim=1)
    wf = w.double()
    if scale is not None:
        wf = (w.float() * scale.float()[:, None, None, None, None]).double()   # kernel scales in fp32
    wq = q(wf.float())
    y = F.conv3d(F.pad(x.double(), (1,) * 6, mode="reflect"), wq.double())
    if shift is not None:
        y = y + shift.double()[None, :, None, None, None]
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = F.leaky_relu(y, slope)
    return y.float()
