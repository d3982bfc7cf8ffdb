"""The projection head of PatchSampleF on the HIP kernels (csrc/amx_mlp.hip): one C-ABI call per head and direction
instead of ~50 small launches (reference: pretraining/models/pretraining_networks.py:338-350 builds the head, :505-511
applies it to the [views * patches, C] samples in train mode)."""
import ctypes
import os

import torch
import torch.nn as nn

from .. import _lib


def parse_head(mlp):
    """nn.Sequential -> (layers, act, slope) with layers = [(Linear, BatchNorm1d), ...], or None when the module is not a
    chain of Linear(bias=False) - BatchNorm1d [- ReLU / LeakyReLU] blocks with the activation everywhere but after the
    last block (the only structure the reference builds)."""
    mods = list(mlp)
    layers, acts = [], []
    i = 0
    while i < len(mods):
        if i + 1 >= len(mods) or not isinstance(mods[i], nn.Linear) or not isinstance(mods[i + 1], nn.BatchNorm1d):
            return None
        lin, bn = mods[i], mods[i + 1]
        if lin.bias is not None or bn.momentum is None or (bn.weight is None) != (bn.bias is None):
            return None
        i += 2
        a = None
        if i < len(mods) and isinstance(mods[i], (nn.ReLU, nn.LeakyReLU)):
            a = mods[i]
            i += 1
        layers.append((lin, bn))
        acts.append(a)
    if not layers or acts[-1] is not None or any(a is None for a in acts[:-1]):
        return None
    kinds = {(type(a), getattr(a, "negative_slope", 0.0)) for a in acts[:-1]}
    if len(kinds) > 1:
        return None
    act, slope = "none", 0.0
    if kinds:
        (t, s), = kinds
        act, slope = ("relu", 0.0) if t is nn.ReLU else ("lrelu", float(s))
    width = layers[0][0].out_features
    if any(l.out_features != width for l, _ in layers) or any(l.in_features != width for l, _ in layers[1:]):
        return None
    if len({bn.eps for _, bn in layers}) > 1 or len({bn.momentum for _, bn in layers}) > 1:
        return None
    return layers, act, slope


def cached_spec(mlp):
    """parse_head once per module object (the structure of a head does not change after PatchSampleF.create_mlp)."""
    spec = mlp.__dict__.get("_amx_head_spec", False)
    if spec is False:
        spec = parse_head(mlp)
        mlp.__dict__["_amx_head_spec"] = spec
    return spec


def unsupported_reason(mlp, x):
    if _lib.exp_env("AMX_NO_MLP_HEAD", "0") == "1":
        return "disabled by AMX_NO_MLP_HEAD=1"
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2:
        return "expects a CUDA fp32 [rows, features] input"
    parsed = cached_spec(mlp)
    if parsed is None:
        return "not a Linear(bias=False) - BatchNorm1d - activation chain"
    layers = parsed[0]
    if not all(bn.training for _, bn in layers):
        return "BatchNorm1d in eval mode (running statistics) is left to the stock modules"
    n, cin = x.shape
    width = layers[0][0].out_features
    if not (1 <= n <= 2048) or cin % 4 or width % 8 or len(layers) > 8 or cin != layers[0][0].in_features:
        return "shape outside the kernel's limits (rows <= 2048, features % 4 == 0, width % 8 == 0)"
    if any(lin.weight.dtype != torch.float32 for lin, _ in layers):
        return "parameters must be fp32"
    return None


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class _MlpHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, spec, *params):
        layers, act, slope = spec
        lib = _lib.load()
        dev = x.device
        L = len(layers)
        n, cin = x.shape
        width = layers[0][0].out_features
        xc = x.detach().contiguous()
        ws = [lin.weight.detach().contiguous() for lin, _ in layers]
        gs = [None if bn.weight is None else bn.weight.detach() for _, bn in layers]
        bs = [None if bn.bias is None else bn.bias.detach() for _, bn in layers]
        rms = [bn.running_mean for _, bn in layers]
        rvs = [bn.running_var for _, bn in layers]
        z = torch.empty((L, n, width), dtype=torch.float32, device=dev)
        y = torch.empty((L, n, width), dtype=torch.float32, device=dev)
        mean = torch.empty((L, width), dtype=torch.float32, device=dev)
        rstd = torch.empty((L, width), dtype=torch.float32, device=dev)
        bn0 = layers[0][1]
        with torch.cuda.device(dev):
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.amx_mlp_head_forward(_lib.ptr(xc), n, cin, width, L, _ptr_array(ws), _ptr_array(gs), _ptr_array(bs),
                                                _ptr_array(rms), _ptr_array(rvs), float(bn0.eps), float(bn0.momentum),
                                                _lib.ACT[act], float(slope), _lib.ptr(z), _lib.ptr(y), _lib.ptr(mean),
                                                _lib.ptr(rstd), st))
        tracked = [bn.num_batches_tracked for _, bn in layers if bn.num_batches_tracked is not None]
        if tracked:
            torch._foreach_add_(tracked, 1)
        ctx.spec, ctx.saved = spec, (xc, ws, gs, z, y, mean, rstd)
        ctx.needs_dx = x.requires_grad
        return y[L - 1]

    @staticmethod
    def backward(ctx, dy):
        layers, act, slope = ctx.spec
        xc, ws, gs, z, y, mean, rstd = ctx.saved
        lib = _lib.load()
        dev = dy.device
        L = len(layers)
        n, cin = xc.shape
        width = ws[0].shape[0]
        dyc = dy.contiguous().float()
        dws = [torch.empty_like(w) for w in ws]
        dgs = [None if g is None else torch.empty_like(g) for g in gs]
        dbs = [None if g is None else torch.empty_like(g) for g in gs]
        dx = torch.empty_like(xc) if ctx.needs_dx else None
        with torch.cuda.device(dev):
            nb = lib.amx_mlp_head_scratch_bytes(n, cin, width)
            sc = torch.empty(nb, dtype=torch.uint8, device=dev)
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.amx_mlp_head_backward(_lib.ptr(dyc), _lib.ptr(xc), n, cin, width, L, _ptr_array(ws), _ptr_array(gs),
                                                 _lib.ACT[act], float(slope), _lib.ptr(z), _lib.ptr(y), _lib.ptr(mean),
                                                 _lib.ptr(rstd), _ptr_array(dws), _ptr_array(dgs), _ptr_array(dbs), _lib.ptr(dx),
                                                 _lib.ptr(sc), nb, st))
        grads = []
        for l in range(L):
            grads.append(dws[l])
            if gs[l] is not None:
                grads += [dgs[l], dbs[l]]
        return (dx, None) + tuple(grads)


class _MlpHeadsFn(torch.autograd.Function):
    """Several heads of one shape class (same rows, width, depth, activation, BatchNorm constants; own input widths) as ONE chain of
    launches (amx_mlp_heads_forward / _backward): the contrastive step's six heads are 6 x ~20 dependent small launches otherwise,
    and a replayed graph pays per dependent node.  Per head the same kernels in the same order: bit-identical to _MlpHeadFn."""

    @staticmethod
    def forward(ctx, specs, nparams, *args):
        nb = len(specs)
        xs = args[:nb]
        lib = _lib.load()
        dev = xs[0].device
        layers0, act, slope = specs[0]
        L = len(layers0)
        n = xs[0].shape[0]
        width = layers0[0][0].out_features
        xcs = [x.detach().contiguous() for x in xs]
        cins = [int(x.shape[1]) for x in xcs]
        ws, gs, bs, rms, rvs = [], [], [], [], []
        for layers, _, _ in specs:
            for lin, bn in layers:
                ws.append(lin.weight.detach().contiguous())
                gs.append(None if bn.weight is None else bn.weight.detach())
                bs.append(None if bn.bias is None else bn.bias.detach())
                rms.append(bn.running_mean)
                rvs.append(bn.running_var)
        z = torch.empty((nb, L, n, width), dtype=torch.float32, device=dev)
        y = torch.empty((nb, L, n, width), dtype=torch.float32, device=dev)
        mean = torch.empty((nb, L, width), dtype=torch.float32, device=dev)
        rstd = torch.empty((nb, L, width), dtype=torch.float32, device=dev)
        bn0 = layers0[0][1]
        cin_arr = (ctypes.c_int * nb)(*cins)
        with torch.cuda.device(dev):
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.amx_mlp_heads_forward(nb, _ptr_array(xcs), n, cin_arr, width, L, _ptr_array(ws), _ptr_array(gs), _ptr_array(bs),
                                                 _ptr_array(rms), _ptr_array(rvs), float(bn0.eps), float(bn0.momentum), _lib.ACT[act],
                                                 float(slope), _ptr_array(list(z)), _ptr_array(list(y)), _ptr_array(list(mean)),
                                                 _ptr_array(list(rstd)), st))
        tracked = [bn.num_batches_tracked for layers, _, _ in specs for _, bn in layers if bn.num_batches_tracked is not None]
        if tracked:
            torch._foreach_add_(tracked, 1)
        ctx.specs, ctx.saved = specs, (xcs, cins, ws, gs, z, y, mean, rstd)
        ctx.needs_dx = [x.requires_grad for x in xs]
        return tuple(y[h, L - 1] for h in range(nb))

    @staticmethod
    def backward(ctx, *dys):
        specs = ctx.specs
        xcs, cins, ws, gs, z, y, mean, rstd = ctx.saved
        lib = _lib.load()
        nb = len(specs)
        layers0, act, slope = specs[0]
        L = len(layers0)
        dev = xcs[0].device
        n = xcs[0].shape[0]
        width = ws[0].shape[0]
        dycs = [(torch.zeros((n, width), dtype=torch.float32, device=dev) if d is None else d.contiguous().float()) for d in dys]
        dws = [torch.empty_like(w) for w in ws]
        dgs = [None if g is None else torch.empty_like(g) for g in gs]
        dbs = [None if g is None else torch.empty_like(g) for g in gs]
        dxs = [torch.empty_like(x) if need else None for x, need in zip(xcs, ctx.needs_dx)]
        cin_arr = (ctypes.c_int * nb)(*cins)
        with torch.cuda.device(dev):
            nbytes = max(lib.amx_mlp_head_scratch_bytes(n, c, width) for c in cins)
            nbytes = (nbytes + 255) // 256 * 256
            sc = torch.empty((nb, nbytes), dtype=torch.uint8, device=dev)
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.amx_mlp_heads_backward(nb, _ptr_array(dycs), _ptr_array(xcs), n, cin_arr, width, L, _ptr_array(ws), _ptr_array(gs),
                                                  _lib.ACT[act], float(slope), _ptr_array(list(z)), _ptr_array(list(y)),
                                                  _ptr_array(list(mean)), _ptr_array(list(rstd)), _ptr_array(dws), _ptr_array(dgs),
                                                  _ptr_array(dbs), _ptr_array(dxs), _ptr_array(list(sc)), nbytes, st))
        grads = []
        for i in range(nb * L):
            grads.append(dws[i])
            if gs[i] is not None:
                grads += [dgs[i], dbs[i]]
        return (None, None) + tuple(dxs) + tuple(grads)


def batchable(mlps, xs):
    """None when ``run_heads`` covers these heads as one batch, else the reason (the caller then runs them one by one)."""
    if not 1 <= len(mlps) <= 8:
        return "1 to 8 heads per batch"
    for mlp, x in zip(mlps, xs):
        why = unsupported_reason(mlp, x)
        if why is not None:
            return why
    specs = [cached_spec(m) for m in mlps]
    l0, act, slope = specs[0]
    key = (len(l0), act, slope, l0[0][0].out_features, l0[0][1].eps, l0[0][1].momentum, xs[0].shape[0],
           tuple(bn.weight is None for _, bn in l0))
    for (layers, a, sl), x in zip(specs, xs):
        if (len(layers), a, sl, layers[0][0].out_features, layers[0][1].eps, layers[0][1].momentum, x.shape[0],
                tuple(bn.weight is None for _, bn in layers)) != key:
            return "heads of different structure"
    return None


def run_heads(mlps, xs):
    """[mlp(x) for mlp, x in zip(mlps, xs)] as one chain of launches; raises with the reason when the batch is not covered."""
    why = batchable(mlps, xs)
    if why is not None:
        raise RuntimeError("projection heads as a batch: " + why)
    specs = tuple(cached_spec(m) for m in mlps)
    params = []
    for layers, _, _ in specs:
        params += head_params(layers)
    return list(_MlpHeadsFn.apply(specs, len(params), *xs, *params))


def head_params(layers):
    ps = []
    for lin, bn in layers:
        ps.append(lin.weight)
        if bn.weight is not None:
            ps += [bn.weight, bn.bias]
    return ps


def run_head(mlp, x):
    """mlp(x) for x [rows, C] on the HIP kernels; raises with the reason when the head is not covered."""
    why = unsupported_reason(mlp, x)
    if why is not None:
        raise RuntimeError("projection head on the HIP kernels: " + why)
    spec = cached_spec(mlp)
    return _MlpHeadFn.apply(x, spec, *head_params(spec[0]))
