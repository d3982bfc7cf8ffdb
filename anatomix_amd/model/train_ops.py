"""Thin torch-tensor wrappers over the training-path entry points of libanatomix_amd.so.

Every function takes / returns CUDA tensors in the library's layout -- activations and gradients dense 16-bit
channels-last ``[N, D, H, W, C]`` (``torch.float16`` or ``torch.bfloat16``), parameters and their gradients fp32 -- and
enqueues on the current stream.  No arithmetic happens in Python; torch only owns the memory.
"""
import ctypes

import torch

from .. import _lib

_PREC = {torch.float16: 0, torch.bfloat16: 1}
ACT = _lib.ACT


def _st(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


_CACHE = {}


def _cached(key, make):
    """Per-process scratch buffers reused across calls (same stream order => no hazards): packed-weight scratch, the
    statistics scratch of the norm kernels and the zero-framed gradient buffers."""
    t = _CACHE.get(key)
    if t is None:
        t = _CACHE[key] = make()
    return t


def _scratch(lib, dev, c):
    nbytes = lib.amx_train_scratch_bytes(2048)
    return _cached(("train_scratch", dev), lambda: torch.empty(nbytes, dtype=torch.uint8, device=dev))


def _as_weight(weight):
    w = weight.detach()
    return w if (w.dtype == torch.float32 and w.is_contiguous()) else w.float().contiguous()


def pack_batch(reqs, dtype, device):
    """One launch for many layers' packed weights.  ``reqs``: list of (weight fp32 contiguous, weight_mode, cin_pad, width): returns one
    uint8 buffer per request (views of one allocation) to hand to ``conv_forward(..., wpk=)``."""
    lib = _lib.load()
    sizes, metas = [], []
    for wt, mode, cin_pad, width in reqs:
        if mode == 0:
            cout_real, cin_real = wt.shape[0], wt.shape[1]
        else:
            cin_real, cout_real = wt.shape[0], wt.shape[1]
        cout = (cout_real + 15) // 16 * 16
        nb = (lib.amx_conv3d_packed_bytes(cin_pad, cout) + 255) // 256 * 256
        sizes.append(nb)
        metas.append((cin_real, cout_real, cout))
    buf = torch.empty(sum(sizes), dtype=torch.uint8, device=device)
    arr = (_lib.PackReq * len(reqs))()
    views, off = [], 0
    for i, ((wt, mode, cin_pad, width), (cin_real, cout_real, cout), nb) in enumerate(zip(reqs, metas, sizes)):
        v = buf[off: off + nb]
        views.append(v)
        arr[i].d_weight, arr[i].d_wpk = wt.data_ptr(), v.data_ptr()
        arr[i].weight_mode, arr[i].cin_real, arr[i].cin_pad = mode, cin_real, cin_pad
        arr[i].cout_real, arr[i].cout, arr[i].w = cout_real, cout, width
        off += nb
    with torch.cuda.device(device):
        _lib.check(lib.amx_conv3d_pack_batch(arr, len(reqs), _PREC[dtype], _st(device)))
    return views


def conv_forward(x0, x1, weight, act="none", slope=0.3, out32=False, shift=None, weight_mode=0, wpk=None):
    """nn.Conv3d(k3, reflect 'same') on cat(x0, nearest_up2(x1)).  weight_mode 0: weight fp32 [Cout, Cin, 3, 3, 3] with Cin <=
    channels of the inputs (the stem's single channel sits in a 16-channel tensor); weight_mode 1: the data-gradient
    convolution of the conv whose FORWARD weight is ``weight`` [Cin_of_x0, Cout_result, 3,3,3] (flip + transpose happen in
    the library's packer).  Returns 16-bit NDHWC, or fp32 NCDHW when ``out32``."""
    lib = _lib.load()
    dev = x0.device
    n, d, h, w, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[-1]
    wt = _as_weight(weight)
    if weight_mode == 0:
        cout_real, cin_real = wt.shape[0], wt.shape[1]
    else:
        cin_real, cout_real = wt.shape[0], wt.shape[1]
    cout = (cout_real + 15) // 16 * 16
    with torch.cuda.device(dev):
        nb = lib.amx_conv3d_packed_bytes(c0 + c1, cout)
        if wpk is not None:                                   # packed by pack_batch for exactly this call
            assert wpk.numel() >= nb
            weight_mode |= _lib.WEIGHTS_PREPACKED
        else:
            wpk = _cached(("wpk", dev, nb), lambda: torch.empty(nb, dtype=torch.uint8, device=dev))
        if out32:
            out = torch.empty((n, cout, d, h, w), dtype=torch.float32, device=dev)
            o16, o32 = None, out
        else:
            out = torch.empty((n, d, h, w, cout), dtype=x0.dtype, device=dev)
            o16, o32 = out, None
        _lib.check(lib.amx_conv3d_k3_reflect_ex(_lib.ptr(x0), c0, _lib.ptr(x1), c1, _lib.ptr(wt), weight_mode, cin_real, cout_real,
                                                None, _lib.ptr(shift), cout, n, d, h, w, ACT[act], slope, _PREC[x0.dtype],
                                                _lib.ptr(wpk), _lib.ptr(o16), _lib.ptr(o32), _st(dev)))
    return out


def bn_train_forward(x, gamma, beta, eps, act="relu", slope=0.3, running_mean=None, running_var=None, momentum=0.1, out=None):
    """BatchNorm3d(train) + activation (one sample at a time: InstanceNorm3d).  Returns (y, save_mean, save_rstd)."""
    lib = _lib.load()
    dev = x.device
    n, c = x.shape[0], x.shape[-1]
    vox = x[0, ..., 0].numel()
    y = torch.empty_like(x) if out is None else out
    mean = torch.empty(c, dtype=torch.float32, device=dev)
    rstd = torch.empty(c, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        sc = _scratch(lib, dev, c)
        _lib.check(lib.amx_bn_train_forward(_lib.ptr(x), _lib.ptr(y), _lib.ptr(gamma), _lib.ptr(beta), float(eps), n, vox, c,
                                            ACT[act], slope, _lib.ptr(sc), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(running_mean),
                                            _lib.ptr(running_var), float(momentum), _PREC[x.dtype], _st(dev)))
    return y, mean, rstd


def new_framed(n, d, h, w, c, dtype, device):
    """Zero-framed gradient buffer [N, D+4, H+4, W+4, C] (frame stays zero; the interior is overwritten by every use)."""
    return torch.zeros((n, d + 4, h + 4, w + 4, c), dtype=dtype, device=device)


def shared_framed(n, d, h, w, c, dtype, device):
    """The framed buffer of this shape shared by all uses in the process: every user overwrites the whole interior and
    nobody writes the frame, so one buffer per shape serves every layer and every step."""
    return _cached(("frame", device, n, d, h, w, c, dtype), lambda: new_framed(n, d, h, w, c, dtype, device))


def interior(framed):
    return framed[:, 2:-2, 2:-2, 2:-2, :]


def bn_act_backward(dy, y, x, mean, rstd, gamma, act="relu", slope=0.3, framed=None, beta=None, recompute=False):
    """Adjoint of bn_train_forward (mean=None: of the bare activation).  Returns (dx_framed, dgamma, dbeta).
    ``recompute`` (needs the norm's ``beta`` as the forward saw it): the activated output ``y`` is not read -- the sign act' needs is
    recomputed from ``x`` (amx_bn_act_backward_recompute), two tensor reads per pass instead of three."""
    lib = _lib.load()
    dev = dy.device
    n, d, h, w, c = dy.shape
    if framed is None:
        framed = new_framed(n, d, h, w, c, dy.dtype, dev)
    dgamma = dbeta = None
    if mean is not None:
        dgamma = torch.empty(c, dtype=torch.float32, device=dev)
        dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        sc = _scratch(lib, dev, c)
        if recompute and mean is not None:
            _lib.check(lib.amx_bn_act_backward_recompute(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma),
                                                         _lib.ptr(beta), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(framed), n, d, h, w,
                                                         c, ACT[act], slope, _lib.ptr(sc), _PREC[dy.dtype], _st(dev)))
        else:
            _lib.check(lib.amx_bn_act_backward(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma),
                                               _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(framed), n, d, h, w, c, ACT[act], slope,
                                               _lib.ptr(sc), _PREC[dy.dtype], _st(dev)))
    return framed, dgamma, dbeta


def conv_dgrad(dx_framed, weight, cin_keep=None, accumulate_into=None):
    """Data gradient of Conv3d(k3, reflect): the forward kernel on the framed output gradient with the flipped, transposed
    weights (packed straight from the forward tensor), then the reflect-padding adjoint.  weight fp32 [Cout, Cin, 3,3,3];
    returns 16-bit [N, D, H, W, Cin_pad16]."""
    return pad_fold(conv_dgrad_framed(dx_framed, weight), accumulate_into)


def conv_dgrad_framed(dx_framed, weight, wpk=None):
    """First half of conv_dgrad: the raw result on the padded domain, 16-bit [N, D+4, H+4, W+4, Cin_pad16]."""
    return conv_forward(dx_framed, None, weight, weight_mode=1, wpk=wpk)


def dgrad_direct_supported(dx_framed, weight):
    """True when ``conv_dgrad_direct`` covers this block (the shapes of the z-march kernels, plain 16-bit storage)."""
    lib = _lib.load()
    n, df, hf, wf, c = dx_framed.shape
    cout = (weight.shape[1] + 15) // 16 * 16
    return bool(lib.amx_conv3d_dgrad_interior_supported(c, cout, df - 4, hf - 4, wf - 4, _PREC[dx_framed.dtype]))


def conv_dgrad_direct(dx_framed, weight, wpk=None):
    """The data gradient [N, D, H, W, Cin_pad16] of the conv whose forward weight is ``weight`` from the zero-framed output gradient, WITHOUT
    computing it on the padded domain: the forward kernel on the interior (halo read from the frame) + the folded shell terms.
    Equals ``pad_fold(conv_dgrad_framed(dx_framed, weight))`` up to the rounding of 16-bit intermediates."""
    lib = _lib.load()
    dev = dx_framed.device
    n, df, hf, wf, c = dx_framed.shape
    d, h, w = df - 4, hf - 4, wf - 4
    wt = _as_weight(weight)
    co_f, ci_f = wt.shape[0], wt.shape[1]
    cout = (ci_f + 15) // 16 * 16
    out = torch.empty((n, d, h, w, cout), dtype=dx_framed.dtype, device=dev)
    with torch.cuda.device(dev):
        nb = lib.amx_conv3d_packed_bytes(c, cout)
        flags = 0
        if wpk is not None:
            assert wpk.numel() >= nb
            flags = _lib.WEIGHTS_PREPACKED
        else:
            wpk = _cached(("wpk", dev, nb), lambda: torch.empty(nb, dtype=torch.uint8, device=dev))
        prec = _PREC[dx_framed.dtype]
        _lib.check(lib.amx_conv3d_dgrad_interior(_lib.ptr(dx_framed), c, _lib.ptr(wt), flags, co_f, ci_f, cout, n, d, h, w, prec,
                                                 _lib.ptr(wpk), _lib.ptr(out), _st(dev)))
        # (one fragment table per call site would be needed if calls of DIFFERENT layers could overlap: they are stream-ordered)
        tab = _cached(("shell_tab", dev), lambda: torch.empty(lib.amx_conv3d_dgrad_shell_scratch_bytes(), dtype=torch.uint8, device=dev))
        _lib.check(lib.amx_conv3d_dgrad_fold_shell(_lib.ptr(dx_framed), c, _lib.ptr(wt), co_f, ci_f, _lib.ptr(out), cout, n, d, h, w, prec,
                                                   _lib.ptr(tab), _st(dev)))
    return out


def pad_fold(g, accumulate_into=None):
    """Second half: the reflect-padding adjoint, [N, D+4, H+4, W+4, C] -> [N, D, H, W, C]."""
    lib = _lib.load()
    dev = g.device
    n, df, hf, wf, cin_pad = g.shape
    d, h, w = df - 4, hf - 4, wf - 4
    din = accumulate_into if accumulate_into is not None else torch.empty((n, d, h, w, cin_pad), dtype=g.dtype, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.amx_pad_fold(_lib.ptr(g), _lib.ptr(din), n, d, h, w, cin_pad, int(accumulate_into is not None),
                                    _PREC[g.dtype], _st(dev)))
    return din


def wgrad_scratch(x0, x1, cout):
    """Allocates (once, on the current stream) the cached partial-sum scratch conv_wgrad uses for this shape."""
    lib = _lib.load()
    n, d, h, w, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[-1]
    with torch.cuda.device(x0.device):
        nbytes = lib.amx_conv3d_wgrad_scratch_bytes(n, d, h, w, cout, c0 + c1)
        return _cached(("wgrad", x0.device, nbytes), lambda: torch.empty(nbytes, dtype=torch.uint8, device=x0.device))


def conv_wgrad(dx_framed, x0, x1, cin_real, cout, out=None, accumulate=False):
    """Weight gradient: fp32 [Cout, cin_real, 3, 3, 3] (``out``: preallocated result, e.g. by the caller's main stream when the
    kernel itself is enqueued on a side stream; ``accumulate``: added to ``out`` instead of overwriting it)."""
    lib = _lib.load()
    dev = x0.device
    n, d, h, w, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[-1]
    dw = out if out is not None else torch.empty((cout, cin_real, 3, 3, 3), dtype=torch.float32, device=dev)
    view = interior(dx_framed)
    es = dx_framed.element_size()
    sn, sz, sy, sx, _ = [s * es for s in view.stride()]
    with torch.cuda.device(dev):
        nbytes = lib.amx_conv3d_wgrad_scratch_bytes(n, d, h, w, cout, c0 + c1)
        sc = _cached(("wgrad", dev, nbytes), lambda: torch.empty(nbytes, dtype=torch.uint8, device=dev))
        _lib.check(lib.amx_conv3d_wgrad(ctypes.c_void_p(view.data_ptr()), sn, sz, sy, sx, _lib.ptr(x0), c0, _lib.ptr(x1), c1,
                                        cin_real, cout, n, d, h, w, _lib.ptr(dw), 1 if accumulate else 0, _lib.ptr(sc), nbytes, _PREC[x0.dtype],
                                        _st(dev)))
    return dw


def pool2(x, mode):
    """nn.MaxPool3d(2) (mode 0) / nn.AvgPool3d(2) (mode 1)."""
    lib = _lib.load()
    n, d, h, w, c = x.shape
    out = torch.empty((n, d // 2, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.amx_pool2(_lib.ptr(x), _lib.ptr(out), n, d // 2, h // 2, w // 2, c, int(mode), _PREC[x.dtype], _st(x.device)))
    return out


def pool2_max(x):
    return pool2(x, 0)


def upsample2_trilinear(x):
    """nn.Upsample(scale_factor=2, mode='trilinear') on 16-bit NDHWC."""
    lib = _lib.load()
    n, d, h, w, c = x.shape
    out = torch.empty((n, 2 * d, 2 * h, 2 * w, c), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.amx_upsample2_trilinear(_lib.ptr(x), _lib.ptr(out), n, d, h, w, c, _PREC[x.dtype], _st(x.device)))
    return out


def upsample2_trilinear_backward(g):
    """Adjoint of upsample2_trilinear: [N, 2D, 2H, 2W, C] -> [N, D, H, W, C]."""
    lib = _lib.load()
    n, d2, h2, w2, c = g.shape
    g = g.contiguous()
    out = torch.empty((n, d2 // 2, h2 // 2, w2 // 2, c), dtype=g.dtype, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(lib.amx_upsample2_trilinear_backward(_lib.ptr(g), _lib.ptr(out), n, d2 // 2, h2 // 2, w2 // 2, c,
                                                        _PREC[g.dtype], _st(g.device)))
    return out


def pool2_max_backward(dp, inp, accumulate_into=None):
    lib = _lib.load()
    n, do, ho, wo, c = dp.shape
    din = accumulate_into if accumulate_into is not None else torch.empty_like(inp)
    with torch.cuda.device(dp.device):
        _lib.check(lib.amx_pool2_max_backward(_lib.ptr(dp), _lib.ptr(inp), _lib.ptr(din), n, do, ho, wo, c,
                                              int(accumulate_into is not None), _PREC[dp.dtype], _st(dp.device)))
    return din


def export_ncdhw(x):
    """16-bit NDHWC -> fp32 NCDHW (a feature tap handed back to torch)."""
    lib = _lib.load()
    n, d, h, w, c = x.shape
    x = x.contiguous()
    out = torch.empty((n, c, d, h, w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.amx_export_ncdhw(_lib.ptr(x), c, n, d, h, w, _lib.ptr(out), _PREC[x.dtype], _st(x.device)))
    return out


def gather_rows(x, coords, channels_last=True):
    """rows [N, P, C] fp32 = x[n, coords[p], :].  ``x``: 16-bit NDHWC [N, D, H, W, C'] (``channels_last``) or fp32 NCDHW [N, C, D, H, W];
    ``coords``: int64 [P, 3] (z, y, x).  Reads the P voxels in place -- no dense fp32 copy of the tensor."""
    lib = _lib.load()
    coords = coords.contiguous()
    p = coords.shape[0]
    if channels_last:
        n, c = x.shape[0], x.shape[4]
        sn, sz, sy, sx, sc = x.stride()
        dtype = _PREC[x.dtype]
    else:
        assert x.dtype == torch.float32
        n, c = x.shape[0], x.shape[1]
        sn, sc, sz, sy, sx = x.stride()
        dtype = 2
    rows = torch.empty((n, p, c), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.amx_gather_rows(ctypes.c_void_p(x.data_ptr()), dtype, sn, sz, sy, sx, sc, _lib.ptr(coords), n, p, c, _lib.ptr(rows),
                                       _st(x.device)))
    return rows


def scatter_rows(rows, coords, dst, accumulate=False):
    """Adjoint of ``gather_rows``: dst[n, coords[p], :C] (= or +=) rows[n, p, :] for a 16-bit NDHWC view ``dst`` (any strides, e.g. the
    interior of a framed buffer); fp32 add, one rounding -- import_ncdhw(accumulate) restricted to the sampled voxels."""
    lib = _lib.load()
    rows = rows.contiguous() if rows.dtype == torch.float32 else rows.float().contiguous()
    coords = coords.contiguous()
    n, p, c = rows.shape
    es = dst.element_size()
    sn, sz, sy, sx, sc = [s * es for s in dst.stride()]
    assert sc == es and dst.shape[0] == n and dst.shape[4] >= c and coords.shape == (p, 3)
    with torch.cuda.device(rows.device):
        _lib.check(lib.amx_scatter_rows(_lib.ptr(rows), _lib.ptr(coords), ctypes.c_void_p(dst.data_ptr()), _PREC[dst.dtype], sn, sz, sy, sx,
                                        n, p, c, int(accumulate), _st(rows.device)))
    return dst


def conv_backward_sampled(grows, coords, x0, weight, cin_real, need_din=True):
    """Backward of a k3 reflect conv (16-bit NDHWC input ``x0``, fp32 ``weight`` [Cout, cin_real, 3, 3, 3]) whose output gradient is the
    sampled rows ``grows`` [N, P, Cout] at ``coords`` [P, 3] and zero elsewhere: (dW fp32, d input 16-bit NDHWC like x0 -- dense, zero
    away from the samples' neighbourhoods).  amx_conv3d_backward_sampled; Cout, cin_real <= 16."""
    lib = _lib.load()
    grows = grows.contiguous() if grows.dtype == torch.float32 else grows.float().contiguous()
    coords = coords.contiguous()
    n, d, h, w, xc = x0.shape
    p, cout = grows.shape[1], grows.shape[2]
    wt = _as_weight(weight)
    dw = torch.empty((cout, cin_real, 3, 3, 3), dtype=torch.float32, device=x0.device)
    din = torch.zeros_like(x0) if need_din else None
    with torch.cuda.device(x0.device):
        nb = lib.amx_conv3d_backward_sampled_scratch_bytes(p)
        sc = torch.empty(nb, dtype=torch.uint8, device=x0.device)
        _lib.check(lib.amx_conv3d_backward_sampled(_lib.ptr(grows), _lib.ptr(coords), _lib.ptr(x0), xc, _lib.ptr(wt), n, p, d, h, w, cout,
                                                   cin_real, _lib.ptr(dw), _lib.ptr(din), xc, _lib.ptr(sc), nb, _PREC[x0.dtype],
                                                   _st(x0.device)))
    return dw, din


def import_input(x, dtype):
    """fp32 [N, Cin <= 16, D, H, W] -> 16-bit channels-last [N, D, H, W, 16] (real channels first, the rest zero) in one pass."""
    lib = _lib.load()
    n, cin, d, h, w = x.shape
    xs = x.detach().contiguous().float()
    out = torch.empty((n, d, h, w, 16), dtype=dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.amx_import_input(_lib.ptr(xs), _lib.ptr(out), n, cin, d, h, w, _PREC[dtype], _st(x.device)))
    return out


def import_ncdhw(g, dst, accumulate=False):
    """fp32 NCDHW gradient -> 16-bit NDHWC view ``dst`` [N, D, H, W, C'] (C' >= C channels per voxel; any strides that are
    multiples of 16 bytes, e.g. the interior of a framed buffer), optionally adding to it."""
    lib = _lib.load()
    g = g.contiguous() if g.dtype == torch.float32 else g.float().contiguous()
    n, c, d, h, w = g.shape
    es = dst.element_size()
    sn, sz, sy, sx, sc = [s * es for s in dst.stride()]
    assert sc == es and tuple(dst.shape[:4]) == (n, d, h, w) and dst.shape[4] >= c
    with torch.cuda.device(g.device):
        _lib.check(lib.amx_import_ncdhw(_lib.ptr(g), ctypes.c_void_p(dst.data_ptr()), n, c, d, h, w, sn, sz, sy, sx,
                                        int(accumulate), _PREC[dst.dtype], _st(g.device)))
    return dst


def upcat_split_backward_framed(g, c0, c1, skip_into=None):
    """pad_fold + upcat_split_backward in one pass over the framed data-gradient result g [N, D+4, H+4, W+4, c0 + c1]."""
    lib = _lib.load()
    n, df, hf, wf, c = g.shape
    d, h, w = df - 4, hf - 4, wf - 4
    assert c == c0 + c1 and g.is_contiguous() and d % 2 == 0 and h % 2 == 0 and w % 2 == 0
    dskip = skip_into if skip_into is not None else torch.empty((n, d, h, w, c0), dtype=g.dtype, device=g.device)
    dlow = torch.empty((n, d // 2, h // 2, w // 2, c1), dtype=g.dtype, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(lib.amx_upcat_split_backward_framed(_lib.ptr(g), _lib.ptr(dskip), _lib.ptr(dlow), n, d // 2, h // 2, w // 2, c0, c1,
                                                       int(skip_into is not None), _PREC[g.dtype], _st(g.device)))
    return dskip, dlow


def upcat_split_backward(dcat, c0, c1, skip_into=None):
    """Adjoint of cat(skip, nearest_up2(low)): dcat [N, D, H, W, c0 + c1(+pad)] -> (dskip [N, D, H, W, c0], dlow [N, D/2, H/2, W/2,
    c1]) in one pass; ``skip_into`` accumulates the skip part into an existing gradient."""
    lib = _lib.load()
    n, d, h, w, c = dcat.shape
    assert c == c0 + c1 and dcat.is_contiguous()
    dskip = skip_into if skip_into is not None else (torch.empty((n, d, h, w, c0), dtype=dcat.dtype, device=dcat.device) if c0 else None)
    dlow = torch.empty((n, d // 2, h // 2, w // 2, c1), dtype=dcat.dtype, device=dcat.device)
    with torch.cuda.device(dcat.device):
        _lib.check(lib.amx_upcat_split_backward(_lib.ptr(dcat), _lib.ptr(dskip), _lib.ptr(dlow), n, d // 2, h // 2, w // 2, c0, c1,
                                                int(skip_into is not None), _PREC[dcat.dtype], _st(dcat.device)))
    return dskip, dlow
