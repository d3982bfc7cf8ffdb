"""Shared helpers for the parity tests (host-side plumbing only)."""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from anatomix_amd import _lib

TORCH_T = {"f16": torch.float16, "bf16": torch.bfloat16, "f16x2": torch.float16, "bf16x2": torch.bfloat16, "strict": torch.bfloat16,
           "f16x2mx": torch.float16}
SPLIT = {"f16x2", "bf16x2", "strict", "f16x2mx"}
MX = {"f16x2mx"}           # AMX_PREC_F16X2_MX: voxel = [hi(C) | lo(C) | per 16-channel chunk: xl8(16) xh8(16)] (include/anatomix_amd.h)


def e4m3(t):
    """OCP e4m3fn rounding (nearest even) with saturation at +-448, as fp32 values."""
    return t.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def e4m3_bytes(t):
    return t.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)


def mx_weight_shift(w):
    """Sw = floor(log2(448 / max|w|)) the way pack_weights_mx_kernel forms it (frexp of the maximum)."""
    mx = float(w.abs().max())
    if not (mx > 0.0):
        return 0
    m, e = np.frexp(np.float32(mx))
    return int((8 if m > 0.875 else 9) - e)


def to_ndhwc_mx(x_ncdhw):
    """NCDHW fp32 -> the row-planar tensor of f16x2mx (include/anatomix_amd.h) as uint8 [N, D, H, 3 C/16, W, 32]: per row the planes
    hi (C/16) | lo (C/16) | e4m3 copies (C/16: [e4m3(2^11 lo) x 16 | e4m3(hi) x 16] per voxel)."""
    xc = x_ncdhw.permute(0, 2, 3, 4, 1).contiguous().float()          # [N, D, H, W, C]
    hi, lo = split_pair(xc, torch.float16)
    n, d, h, w, c = xc.shape
    assert c % 16 == 0
    k = c // 16
    plane = lambda t: t.reshape(n, d, h, w, k, 16).permute(0, 1, 2, 4, 3, 5)     # [N, D, H, k, W, 16]
    ph = plane(hi).contiguous().view(torch.uint8).reshape(n, d, h, k, w, 32)
    pl = plane(lo).contiguous().view(torch.uint8).reshape(n, d, h, k, w, 32)
    px = torch.cat((plane(e4m3_bytes(lo.float() * 2048.0)), plane(e4m3_bytes(hi.float()))), dim=-1)
    return torch.cat((ph, pl, px), dim=3).contiguous()


def from_ndhwc_mx(raw, c, parts=False):
    """uint8 [N, D, H, 3 C/16, W, 32] -> (NCDHW fp32 value hi + lo, the copy planes [N, D, H, C/16, W, 32][, stored hi, lo as [N, D, H, W, C]])."""
    k = c // 16
    n, d, h, _, w, _ = raw.shape
    unplane = lambda t: t.contiguous().view(torch.float16).reshape(n, d, h, k, w, 16).permute(0, 1, 2, 4, 3, 5).reshape(n, d, h, w, c)
    hi, lo = unplane(raw[:, :, :, :k]), unplane(raw[:, :, :, k:2 * k])
    val = (hi.float() + lo.float()).permute(0, 4, 1, 2, 3).contiguous()
    x8 = raw[:, :, :, 2 * k:]
    return (val, x8, hi, lo) if parts else (val, x8)


def split_pair(x, dtype):
    """fp32 -> (hi, lo) of the strict precisions: hi = round(x), lo = round(x - hi)."""
    hi = x.to(dtype)
    lo = (x - hi.float()).to(dtype)
    return hi, lo


def to_ndhwc_split(x_ncdhw, dtype):
    """NCDHW fp32 -> channels-last [N, D, H, W, 2C] with the voxel layout [hi(C) | lo(C)]."""
    hi, lo = split_pair(x_ncdhw.permute(0, 2, 3, 4, 1).contiguous().float(), dtype)
    return torch.cat((hi, lo), dim=-1).contiguous()


def from_ndhwc_split(x):
    """[N, D, H, W, 2C] split storage -> NCDHW fp32 (hi + lo)."""
    c = x.shape[-1] // 2
    return (x[..., :c].float() + x[..., c:].float()).permute(0, 4, 1, 2, 3).contiguous()


def q_storage(t, precision):
    """The value a tensor takes when stored in `precision` (one 16-bit rounding, or the hi + lo pair)."""
    dt = TORCH_T[precision]
    if precision in SPLIT:
        hi, lo = split_pair(t.float(), dt)
        return hi.float() + lo.float()
    return t.to(dt).float()


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


def run_conv(device, x0, x1, w, scale, shift, act, precision, planar=False, slope=0.3, no_scratch=False):
    """Calls amx_conv3d_k3_reflect.  x0/x1: NCDHW float CPU tensors (x1 half resolution or None)."""
    lib = _lib.load()
    tdt = TORCH_T[precision]
    split = precision in SPLIT
    n, c0, d, h, ww = x0.shape
    cout = w.shape[0]
    c1 = 0 if x1 is None else x1.shape[1]
    mx = precision in MX
    pack = to_ndhwc_mx if mx else ((lambda t: to_ndhwc_split(t, tdt)) if split else (lambda t: to_ndhwc(t, tdt)))
    dx0 = pack(x0).to(device)
    dx1 = None if x1 is None else pack(x1).to(device)
    dw = w.reshape(cout, c0 + c1, 27).contiguous().float().to(device)
    dsc = None if scale is None else scale.float().to(device)
    dsh = None if shift is None else shift.float().to(device)
    wpk = torch.empty(lib.amx_conv3d_packed_bytes(c0 + c1, cout), dtype=torch.uint8, device=device)
    if planar:
        out = torch.full((n, cout, d, h, ww), float("nan"), dtype=torch.float32, device=device)
        o16, o32 = None, out
    else:
        out = torch.full((n, d, h, ww, cout * (3 if mx else (2 if split else 1))), float("nan"), dtype=tdt, device=device)
        if mx:
            out = torch.full((n, d, h, 3 * cout // 16, ww, 32), 0x7f, dtype=torch.uint8, device=device)
        o16, o32 = out, None
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    # split-K scratch (conv3d_k3_ks, deep levels with few voxels): offered whenever the shape asks for it, poisoned first
    sb = 0 if no_scratch else lib.amx_conv3d_scratch_bytes(c0, c1, cout, n, d, h, ww, _lib.PRECISION[precision])
    scratch = torch.full((max(sb, 4) // 4,), float("nan"), dtype=torch.float32, device=device) if sb else None
    _lib.check(lib.amx_conv3d_k3_reflect_ws(_lib.ptr(dx0), c0, _lib.ptr(dx1), c1, _lib.ptr(dw), _lib.ptr(dsc),
                                            _lib.ptr(dsh), cout, n, d, h, ww, act, slope, _lib.PRECISION[precision],
                                            _lib.ptr(wpk), _lib.ptr(o16), _lib.ptr(o32), _lib.ptr(scratch), sb, st))
    torch.cuda.synchronize(device)
    out = out.cpu()
    if planar:
        return out
    if mx:                    # the conv writes the pair only (its consumers -- norm apply, pool, upsample -- write the copies)
        return from_ndhwc_mx(out, cout)[0]
    return from_ndhwc_split(out) if split else from_ndhwc(out.float())


def ref_conv_mx(x0, x1, w, scale, shift, act, slope=0.3):
    """The arithmetic of AMX_PREC_F16X2_MX restated on the CPU (fp64 sums): Wh * xh exactly as the f16 MFMA multiplies it, plus the
    two correction products from e4m3 copies under one power-of-two scale."""
    def parts(t):
        hi, lo = split_pair(t.float(), torch.float16)
        return hi.float(), lo.float()
    xh, xl = parts(x0)
    if x1 is not None:
        uh, ul = parts(x1)
        up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
        xh, xl = torch.cat((xh, up(uh)), 1), torch.cat((xl, up(ul)), 1)
    wf = w.float() if scale is None else w.float() * scale.float()[:, None, None, None, None]
    wh = wf.to(torch.float16).float()
    wl = wf - wh
    sw = mx_weight_shift(wf)
    conv = lambda a, b: F.conv3d(F.pad(a.double(), (1,) * 6, mode="reflect"), b.double())
    y = conv(xh, wh) + (conv(e4m3(xl * 2048.0), e4m3(wh * 2.0 ** sw)) + conv(e4m3(xh), e4m3(wl * 2.0 ** (sw + 11)))) * 2.0 ** (-(sw + 11))
    if shift is not None:
        y = y + shift.double()[None, :, None, None, None]
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = F.leaky_relu(y, slope)
    return y.float()


def ref_conv(x0, x1, w, scale, shift, act, precision, slope=0.3):
    """CPU reference with the SAME input/weight rounding as the kernel, fp32 accumulation."""
    q = lambda t: q_storage(t, precision)
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


def ref_conv_fp64(x0, x1, w, scale, shift, act, slope=0.3):
    """The convolution on the un-rounded fp32 operands, accumulated in fp64."""
    x = x0.double()
    if x1 is not None:
        x = torch.cat((x, F.interpolate(x1.double(), scale_factor=2, mode="nearest")), dim=1)
    wf = w.double()
    if scale is not None:
        wf = wf * scale.double()[:, None, None, None, None]
    y = F.conv3d(F.pad(x, (1,) * 6, mode="reflect"), wf)
    if shift is not None:
        y = y + shift.double()[None, :, None, None, None]
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = F.leaky_relu(y, slope)
    return y.float()


def run_conv_merged(device, x0, x1, w, scale, shift, act, precision, slope=0.3):
    """Calls amx_conv3d_upcat_merged (skip conv into a scratch tensor + merged-tap launch over the upsampled channels that adds it)."""
    lib = _lib.load()
    tdt = TORCH_T[precision]
    split = precision in SPLIT
    n, c0, d, h, ww = x0.shape
    cout, c1 = w.shape[0], x1.shape[1]
    pack = (lambda t: to_ndhwc_split(t, tdt)) if split else (lambda t: to_ndhwc(t, tdt))
    dx0, dx1 = pack(x0).to(device), pack(x1).to(device)
    dw = w.reshape(cout, c0 + c1, 27).contiguous().float().to(device)
    dsc = None if scale is None else scale.float().to(device)
    dsh = None if shift is None else shift.float().to(device)
    wpk = torch.empty(lib.amx_conv3d_upcat_merged_packed_bytes(c0, c1, cout), dtype=torch.uint8, device=device)
    k = 2 if split else 1
    part = torch.full((n * d * h * ww * cout * k,), float("nan"), dtype=tdt, device=device)
    out = torch.full((n, d, h, ww, cout * k), float("nan"), dtype=tdt, device=device)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _lib.check(lib.amx_conv3d_upcat_merged(_lib.ptr(dx0), c0, _lib.ptr(dx1), c1, _lib.ptr(dw), _lib.ptr(dsc), _lib.ptr(dsh), cout,
                                           n, d, h, ww, act, slope, _lib.PRECISION[precision], _lib.ptr(wpk), _lib.ptr(part),
                                           _lib.ptr(out), st))
    torch.cuda.synchronize(device)
    out = out.cpu()
    return from_ndhwc_split(out) if split else from_ndhwc(out.float())


def ref_conv_upcat_merged(x0, x1, w, scale, shift, act, precision, slope=0.3, round_partial=False):
    """Reference for the merged-tap kernel (amx_conv3d_upcat.hip): the skip segment is a plain
    reflect-padded 3x3x3 convolution with rounded weights; the nearest-upsampled segment is, per
    output parity class, a 2x2x2 convolution over the REPLICATE-padded low-res tensor whose weights
    are the fp32 sums of the original taps that hit the same low-res voxel, rounded once."""
    tdt = TORCH_T[precision]
    q = lambda t: t.to(tdt).float()
    c0 = x0.shape[1]
    wf = w.float()
    sc = torch.ones(w.shape[0]) if scale is None else scale.float()
    xs = q(x0).double()
    y = F.conv3d(F.pad(xs, (1,) * 6, mode="reflect"), q(wf[:, :c0] * sc[:, None, None, None, None]).double())
    if round_partial:                                   # two-launch form: the skip partial sums are stored in the 16-bit storage type
        y = q(y.float()).double()
    part = torch.zeros_like(y)                          # the merged-tap part (its own launch in amx_conv3d_upmerge.hip)
    lo = F.pad(q(x1).double(), (1,) * 6, mode="replicate")
    wu = wf[:, c0:]                                     # merged in fp32 first, gain applied to the sum (kernel order)
    sets = {0: [[0], [1, 2]], 1: [[0, 1], [2]]}          # parity -> taps merged into low offset e = 0, 1
    n, _, d, h, ww = x0.shape
    for pz in (0, 1):
        for py in (0, 1):
            for px in (0, 1):
                wm = torch.zeros(w.shape[0], wu.shape[1], 2, 2, 2)
                for ez in (0, 1):
                    for ey in (0, 1):
                        for ex in (0, 1):
                            acc = torch.zeros(w.shape[0], wu.shape[1])
                            for kz in sets[pz][ez]:
                                for ky in sets[py][ey]:
                                    for kx in sets[px][ex]:
                                        acc = acc + wu[:, :, kz, ky, kx]
                            wm[:, :, ez, ey, ex] = acc
                sub = lo[:, :, pz:pz + d // 2 + 1, py:py + h // 2 + 1, px:px + ww // 2 + 1]
                part[:, :, pz::2, py::2, px::2] = F.conv3d(sub, q(wm * sc[:, None, None, None, None]).double())
    y = y + part
    if shift is not None:
        y = y + shift.double()[None, :, None, None, None]
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = F.leaky_relu(y, slope)
    return y.float()
