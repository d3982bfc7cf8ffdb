"""Differentiable train-mode forward of ``Unet`` on the HIP kernels (the UNet inside the reference's contrastive
step, pretraining/models/supcl_model.py:603-661, 735-742: ``netG(reals, nce_layers, False)`` with BatchNorm in train
mode, then ``loss.backward()``).

One ``torch.autograd.Function`` covers the whole network: the forward runs conv -> train-mode BatchNorm -> activation
blocks, max-pools and the upsample + concat convs through the library's single-operator entry points
(``anatomix_amd.model.train_ops``), keeps what the adjoint needs (raw conv outputs, activations, batch statistics) in
16-bit channels-last tensors, and the backward walks the blocks in reverse: activation/BatchNorm adjoint -> weight
gradient (MFMA kernel) -> data gradient (forward kernel on the zero-framed gradient with flipped weights + reflect fold)
-> max-pool / upsample / concat adjoints.  Storage precision is ``model.train_precision`` (``model.precision`` when set explicitly, else "bf16": "bf16" mirrors the reference's
bf16 autocast); parameter gradients and BatchNorm statistics are fp32.

Supported configuration: norm 'batch' (batch statistics + running-stat update; layers in eval mode use their frozen
running statistics, folded into the conv), 'instance' / 'instance_affine' (the same kernels, one sample at a time), conv bias, activation relu/lrelu, pooling 'Max' / 'Avg', interp 'nearest' (fused
into the concat convs) / 'trilinear' (materialised + its adjoint kernel), doubleconv either; feature taps at conv /
norm / activation / pool / upsample ids and at the output conv.  Everything else raises (the caller can still opt into the stock-module
path).
"""
import os

import torch
import torch.nn as nn

from .. import _lib
from . import train_ops as T

_DT = {"f16": torch.float16, "fp16": torch.float16, "float16": torch.float16, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}


OVERLAP_WGRAD = _lib.exp_env("AMX_NO_OVERLAP_WGRAD", "0") != "1"   # weight gradients on a side stream, beside the data gradient of the same block
OVERLAP_WGRAD_MIN_W = int(_lib.exp_env("AMX_OVERLAP_WGRAD_MIN_W", "32"))   # ... of blocks at least this wide (narrower: the two joins cost what the overlap returns; 6.41 -> 6.37 ms)
RECOMPUTE_ACT = _lib.exp_env("AMX_BN_BWD_RECOMPUTE", "1") != "0"      # norm adjoint: sign of the activation's argument from x, y not read
SPLIT_CONCAT_DGRAD = int(_lib.exp_env("AMX_SPLIT_CONCAT_DGRAD", "1"))      # 1: the 48 -> 16 layer's data gradient as two z-march launches; 2: every concat layer
FUSED_FOLD_SPLIT = _lib.exp_env("AMX_FUSED_FOLD_SPLIT", "1") != "0"   # concat layers: pad_fold + channel split + child sum in one pass
SPARSE_OUTPUT_TAP = _lib.exp_env("AMX_DENSE_OUTPUT_TAP", "0") != "1"   # output conv tapped at sampled voxels only: its backward from the rows
DIRECT_DGRAD = _lib.exp_env("AMX_NO_DIRECT_DGRAD", "0") != "1"   # data gradient: interior launch + shell terms instead of the framed domain + pad_fold
PACK_ASIDE = _lib.exp_env("AMX_NO_PACK_ASIDE", "0") != "1"     # A/B: both passes' weight packing on a side stream at the start of the forward
BATCH_PACK = _lib.exp_env("AMX_NO_BATCH_PACK", "0") != "1"     # packed weights of a pass in one launch (T.pack_batch)
_SIDE = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device)
    return _SIDE[key]


def unsupported_reason(model, x, layers):
    c = model._cfg
    if model.train_precision not in _DT:
        return (f"the HIP training path stores activations in f16 / bf16 (the reference trains under bf16 autocast); "
                f"precision '{model.train_precision}' is an inference mode")
    if c["dimension"] != 3 or c["pad_type"] != "reflect" or c["residual_connection"]:
        return "only dimension=3, pad_type='reflect', residual_connection=False are implemented"
    if c["norm"] not in ("batch", "instance", "instance_affine") or c["activation"] not in ("relu", "lrelu") or c["final_act"] != "none":
        return "the HIP training path covers norm batch / instance / instance_affine, activation relu/lrelu, final_act='none'"
    if c["pooling"] not in ("Max", "Avg") or c["interp"] not in ("nearest", "trilinear"):
        return "the HIP training path covers pooling Max / Avg, interp nearest / trilinear"
    if c["input_nc"] != 1 or c["ngf"] % 16 or c["output_nc"] % 16 or c["output_nc"] > 32:
        return "the HIP training path needs input_nc == 1, ngf a multiple of 16, output_nc in {16, 32}"
    if x.dim() != 5 or x.shape[1] != 1 or not x.is_cuda:
        return "expected a CUDA input of shape [N, 1, D, H, W]"
    m = 1 << c["num_downs"]
    if any(s % m or (s >> c["num_downs"]) < 2 for s in x.shape[2:]) or x.shape[4] < 32 or x.shape[4] > 128:
        return "spatial dims must be divisible by 2^num_downs, >= 2 at the bottleneck, 32 <= W <= 128"
    kinds = _module_kinds(model)
    for l in layers:
        if not (0 <= l < len(kinds)) or kinds[l] not in ("conv", "norm", "act", "pool", "up"):
            return "feature taps are implemented at conv / norm / activation / pool / upsample ids"
        if kinds[l] == "up" and not c["use_skip_connection"]:
            return "a tap at an upsample id without skip connections is not implemented in the HIP training path"
    return None


def _bn_momentum(bn):
    """torch: momentum=None means a cumulative moving average, factor 1 / num_batches_tracked (counted including this batch)."""
    if bn.momentum is not None:
        return bn.momentum
    if torch.cuda.is_current_stream_capturing():
        # the factor 1 / (n + 1) is read back from the device (a host synchronisation, illegal during capture) and would be frozen
        # into the graph for every replay; the reference never builds its norms this way (network.py:148-152: default momentum)
        raise NotImplementedError("BatchNorm3d(momentum=None) (cumulative moving average) cannot be captured in a HIP graph: "
                                  "run the step eagerly or give the layer a momentum")
    if bn.num_batches_tracked is None:
        raise NotImplementedError("BatchNorm3d(momentum=None) without num_batches_tracked in the HIP training path")
    return 1.0 / (int(bn.num_batches_tracked.item()) + 1)


def _module_kinds(model):
    kinds = []
    for mod in model.model:
        if isinstance(mod, nn.Conv3d):
            kinds.append("conv")
        elif isinstance(mod, (nn.BatchNorm3d, nn.InstanceNorm3d)):
            kinds.append("norm")
        elif isinstance(mod, (nn.ReLU, nn.LeakyReLU)):
            kinds.append("act")
        elif isinstance(mod, (nn.MaxPool3d, nn.AvgPool3d)):
            kinds.append("pool")
        elif isinstance(mod, nn.Upsample):
            kinds.append("up")
        else:
            kinds.append("other")
    return kinds


def _to_cl(t, dt):            # fp32 NCDHW -> 16-bit NDHWC
    return t.permute(0, 2, 3, 4, 1).to(dt).contiguous()


def _to_ncdhw(t):             # 16-bit NDHWC -> fp32 NCDHW
    return T.export_ncdhw(t) if t.shape[-1] % 8 == 0 else t.permute(0, 4, 1, 2, 3).float().contiguous()


class _UnetTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, layers, sampler, *params):
        # sampler (None: dense taps): callable (module id, (d, h, w)) -> int64 coords [P, 3]; the taps then come back as the P sampled
        # rows [N, P, C] fp32 of each tapped tensor (gathered in place from the 16-bit channels-last storage) instead of dense fp32
        # NCDHW copies, and the backward scatters the row gradients straight into the framed gradient buffers
        dt = _DT[model.train_precision]
        coords_of = {}
        # hook of the sampled route (an attribute of the sampler): on_start() once the first block's kernels are enqueued -- work that
        # does not depend on the forward (the coordinate draws) is enqueued behind them instead of in front of the whole forward
        on_start = getattr(sampler, "on_start", None)
        # an output nobody differentiates (the network output next to sampled taps: 268 MB at 128^3) arrives as None in backward, not as a
        # dense zero tensor that would then be imported
        ctx.set_materialize_grads(False)
        act = model._cfg["activation"]
        trilinear = model._cfg["interp"] == "trilinear"
        kinds = _module_kinds(model)
        mods = list(model.model)
        dev = x.device
        n, _, d, h, w = x.shape
        # Packed weights of every plain conv in ONE launch per pass (T.pack_batch) instead of one small launch inside each conv call.  Which
        # convs, with how many stored input channels and at which width, is recorded by the first forward / backward of a given input
        # shape (which pack per call) and replayed afterwards; a conv whose recorded shape does not match packs itself as before.
        # While a HIP graph is being captured both launches go to a side stream here, beside the input import -- the backward's packing
        # (the weights do not change in between) is then off the main stream altogether.
        pkey = (tuple(x.shape), dt)
        plan = getattr(model, "_pack_plan", {}).get(pkey) if BATCH_PACK else None
        bplan0 = getattr(model, "_pack_plan_bwd", {}).get(pkey) if BATCH_PACK else None
        packs, rec = {}, {}
        ctx.bpacks_pre = None
        pack_side = None
        if plan:
            ids = sorted(plan)
            if x.is_cuda and PACK_ASIDE and torch.cuda.is_current_stream_capturing():
                pack_side = _side_stream(dev)
                main_s = torch.cuda.current_stream(dev)
                pack_side.wait_stream(main_s)
                with torch.cuda.stream(pack_side):
                    # (the requests are formed ON the side stream: a parameter that is not fp32-contiguous gets a temporary copy there,
                    #  which the allocator then only recycles behind the side stream's pack kernel)
                    reqs = [(T._as_weight(mods[j].weight), 0, plan[j][0], plan[j][1]) for j in ids]
                    views = T.pack_batch(reqs, dt, dev)
                    for v in views:
                        v.record_stream(main_s)
                    if bplan0:
                        bids = sorted(bplan0)
                        bviews = T.pack_batch([(T._as_weight(mods[j].weight), 1, bplan0[j][0], bplan0[j][1]) for j in bids], dt, dev)
                        for v in bviews:
                            v.record_stream(main_s)
                        ctx.bpacks_pre = (dict(bplan0), dict(zip(bids, bviews)))
            else:
                reqs = [(T._as_weight(mods[j].weight), 0, plan[j][0], plan[j][1]) for j in ids]
                views = T.pack_batch(reqs, dt, dev)
            packs = dict(zip(ids, views))
        # the single input channel, padded to one MFMA chunk: one pass (zero fill + cast + strided copy were three, 56 us at 128^3 x 2)
        xin = T.import_input(x[:, :1], dt) if x.is_cuda else None
        if pack_side is not None:
            torch.cuda.current_stream(dev).wait_stream(pack_side)
        if xin is None:
            xin = torch.zeros((n, d, h, w, 16), dtype=dt, device=dev)
            xin[..., 0] = x.detach()[:, 0].to(dt)
        tensors = {"x": xin}
        blocks, ops, skips = [], [], []
        cur, pending_low = "x", None
        taps = {}
        up_taps = {}                                                     # upsample id -> (skip tensor, low-resolution tensor)
        tracked = []                                                     # num_batches_tracked of every BatchNorm: one foreach add

        def packed(j, cin_pad, width, cout):
            if (cin_pad, cout) == (48, 16):                              # the 16 + up32 -> 16 merged-tap layer packs its own format
                return None
            rec[j] = (cin_pad, width)
            return packs.get(j) if plan and plan.get(j) == (cin_pad, width) else None

        i = 0
        while i < len(mods):
            k = kinds[i]
            if k == "conv":
                conv = mods[i]
                has_bn = i + 1 < len(mods) and kinds[i + 1] == "norm"
                has_act = i + 1 + int(has_bn) < len(mods) and kinds[i + 1 + int(has_bn)] == "act"
                in0, in1 = (skips.pop(), pending_low) if (pending_low is not None and model.use_skip_connection) else (cur, None)
                if pending_low is not None and not model.use_skip_connection:
                    raise NotImplementedError("upsample without skip connection in the HIP training path")
                pending_low = None
                cat_parts = None
                if in1 is not None and trilinear:
                    # trilinear: the upsampled tensor is materialised and concatenated (the nearest case is fused into the conv)
                    name_cat = f"cat{i}"
                    tensors[name_cat] = torch.cat([tensors[in0], T.upsample2_trilinear(tensors[in1])], dim=-1)
                    cat_parts, in0, in1 = (in0, in1), name_cat, None
                bias = None if conv.bias is None else conv.bias.detach().float().contiguous()
                blk = dict(idx=i, conv=conv, in0=in0, in1=in1, cin=conv.in_channels, cout=conv.out_channels, name=f"y{i}",
                           alias_ids=[i + 1 + a for a in range(int(has_bn) + int(has_act))], cat_parts=cat_parts)
                if has_bn and isinstance(mods[i + 1], nn.BatchNorm3d) and not mods[i + 1].training:
                    # BatchNorm with frozen statistics (the reference freezes single layers, pretraining/models/
                    # base_model.py:175-184; also a whole network in eval mode under autograd): y = act(a * conv(x) + b) with
                    # a, b from the running statistics -- folded into the conv's weights and shift, no norm kernels at all
                    bn = mods[i + 1]
                    gam = None if bn.weight is None else bn.weight.detach().float()
                    bet = None if bn.bias is None else bn.bias.detach().float()
                    a = (bn.running_var.float() + bn.eps).rsqrt()
                    if gam is not None:
                        a = a * gam
                    b = -bn.running_mean.float() * a
                    if bet is not None:
                        b = b + bet
                    if bias is not None:
                        b = b + bias * a
                    Y = T.conv_forward(tensors[in0], None if in1 is None else tensors[in1],
                                       conv.weight.detach().float() * a.view(-1, 1, 1, 1, 1), act if has_act else "none", 0.3,
                                       shift=b.contiguous())
                    blk.update(bn=bn, frozen=True, a=a, Y=Y, act=act if has_act else "none")
                    tensors[blk["name"]] = Y
                    if sampler is not None and (i in layers or any(j in layers for j in blk["alias_ids"])):
                        raise NotImplementedError("sampled taps at a frozen-statistics block")
                    if i in layers:                                      # pre-norm tap: the raw convolution, computed only when asked for
                        taps[i] = _to_ncdhw(T.conv_forward(tensors[in0], None if in1 is None else tensors[in1], conv.weight, shift=bias))
                    for j in blk["alias_ids"]:
                        if j in layers:
                            taps[j] = _to_ncdhw(Y)
                    i += 1 + int(has_act)
                elif has_bn:
                    bn = mods[i + 1]
                    cin_pad = tensors[in0].shape[-1] + (0 if in1 is None else tensors[in1].shape[-1])
                    X = T.conv_forward(tensors[in0], None if in1 is None else tensors[in1], conv.weight, shift=bias,
                                       wpk=packed(i, cin_pad, tensors[in0].shape[3], (conv.out_channels + 15) // 16 * 16))
                    gam = None if bn.weight is None else bn.weight.detach()
                    bet = None if bn.bias is None else bn.bias.detach()
                    if isinstance(bn, nn.BatchNorm3d):
                        Y, mean, rstd = T.bn_train_forward(X, gam, bet, bn.eps, act if has_act else "none", 0.3,
                                                           bn.running_mean, bn.running_var,
                                                           _bn_momentum(bn))
                        if bn.num_batches_tracked is not None:
                            tracked.append(bn.num_batches_tracked)
                    else:
                        # InstanceNorm3d: the same statistics kernels over one sample at a time, no running statistics
                        Y = torch.empty_like(X)
                        stats = [T.bn_train_forward(X[s:s + 1], gam, bet, bn.eps, act if has_act else "none", 0.3, out=Y[s:s + 1])[1:]
                                 for s in range(X.shape[0])]
                        mean = torch.stack([m for m, _ in stats])
                        rstd = torch.stack([r for _, r in stats])
                    blk.update(bn=bn, X=X, Y=Y, mean=mean, rstd=rstd, act=act if has_act else "none")
                    tensors[blk["name"]] = Y
                    if i in layers:
                        if sampler is not None:
                            coords_of[i] = sampler(i, tuple(X.shape[1:4]))
                            taps[i] = T.gather_rows(X, coords_of[i])[..., : blk["cout"]]
                        else:
                            taps[i] = _to_ncdhw(X)                       # pre-norm conv output
                    for j in blk["alias_ids"]:
                        if j in layers:
                            if sampler is not None:
                                raise NotImplementedError("sampled taps are implemented at conv ids (pre-norm outputs) and the output conv")
                            taps[j] = _to_ncdhw(Y)                       # in-place activation aliases the norm output
                    i += 1 + int(has_act)
                else:                                                    # the bare output conv
                    out = T.conv_forward(tensors[in0], None, conv.weight, out32=True, shift=bias,
                                         wpk=packed(i, tensors[in0].shape[-1], tensors[in0].shape[3], (conv.out_channels + 15) // 16 * 16))
                    blk.update(bn=None, final=True)
                    tensors[blk["name"]] = out
                    if sampler is not None and i in layers:
                        coords_of[i] = sampler(i, tuple(out.shape[2:]))
                        taps[i] = T.gather_rows(out, coords_of[i], channels_last=False)
                blocks.append(blk)
                ops.append(("conv", blk))
                if on_start is not None:
                    on_start()
                    on_start = None
                cur = blk["name"]
                if model.use_skip_connection and i in model.encoder_idx:
                    skips.append(cur)
            elif k == "pool":
                dst = f"p{i}"
                avg = isinstance(mods[i], nn.AvgPool3d)
                tensors[dst] = T.pool2(tensors[cur], 1 if avg else 0)
                ops.append(("pool", cur, dst, avg, i))
                cur = dst
                if i in layers:
                    if sampler is not None:
                        raise NotImplementedError("sampled taps are implemented at conv ids (pre-norm outputs) and the output conv")
                    taps[i] = _to_ncdhw(tensors[dst])
            elif k == "up":
                pending_low = cur
                if i in layers and sampler is not None:
                    raise NotImplementedError("sampled taps are implemented at conv ids (pre-norm outputs) and the output conv")
                if i in layers:
                    # the reference takes this tap AFTER torch.cat((skip, upsampled), 1) (network.py:500-502): materialised
                    # only when asked for -- the convolution that follows still reads skip and low-resolution tensor directly
                    low = tensors[cur]
                    up = T.upsample2_trilinear(low) if trilinear else \
                        low[:, :, None, :, None, :, None, :].expand(low.shape[0], low.shape[1], 2, low.shape[2], 2, low.shape[3], 2,
                                                                     low.shape[4]).reshape(low.shape[0], 2 * low.shape[1],
                                                                                           2 * low.shape[2], 2 * low.shape[3],
                                                                                           low.shape[4])
                    taps[i] = _to_ncdhw(torch.cat([tensors[skips[-1]], up], dim=-1))
                    up_taps[i] = (skips[-1], cur)
            else:
                raise NotImplementedError(f"module {i} ({type(mods[i]).__name__}) in the HIP training path")
            i += 1
        if BATCH_PACK and not plan:
            model.__dict__.setdefault("_pack_plan", {})[pkey] = rec
        ctx.pkey = pkey
        if tracked:
            torch._foreach_add_(tracked, 1)
        ctx.coords_of = coords_of
        ctx.up_taps, ctx.trilinear = up_taps, trilinear
        ctx.model, ctx.tensors, ctx.ops, ctx.layers, ctx.dt = model, tensors, ops, sorted(taps), dt
        ctx.param_ids = [id(p) for p in model.parameters()]
        ctx.final_idx = blocks[-1]["idx"]
        # popped, not read: the backward never needs the network's output, and an OUTPUT kept in ctx is a reference cycle (output ->
        # grad_fn -> ctx -> output) that only the garbage collector frees -- until then the parameters' AccumulateGrad nodes stay bound
        # to the stream of this call, and a HIP-graph capture of the same modules would run them there (outside the capture)
        out = tensors.pop(blocks[-1]["name"])
        return (out,) + tuple(taps[l] for l in sorted(taps))

    @staticmethod
    def backward(ctx, dout, *dtaps):
        model, tensors, dt = ctx.model, ctx.tensors, ctx.dt
        dtap = {l: g for l, g in zip(ctx.layers, dtaps) if g is not None}
        grads, pgrads, frames = {}, {}, {}
        dx_in = None

        def add_grad(name, g):
            grads[name] = g if name not in grads else grads[name] + g

        def frame(shape, c):
            key = (tuple(shape), c)
            if key not in frames:
                frames[key] = T.shared_framed(shape[0], shape[1], shape[2], shape[3], c, dt, tensors["x"].device)
            return frames[key]

        # gradients of taps taken at upsample ids: split the concatenated gradient, skip part as is, upsampled part through the
        # adjoint of the interpolation
        for uid, (skip_name, low_name) in ctx.up_taps.items():
            if uid not in dtap:
                continue
            g = dtap.pop(uid)
            cs = tensors[skip_name].shape[-1]
            gs = torch.empty_like(tensors[skip_name])
            add_grad(skip_name, T.import_ncdhw(g[:, :cs], gs))
            low = tensors[low_name]
            gu = torch.empty((low.shape[0], 2 * low.shape[1], 2 * low.shape[2], 2 * low.shape[3], low.shape[4]), dtype=dt,
                             device=low.device)
            T.import_ncdhw(g[:, cs:], gu)
            if ctx.trilinear:
                add_grad(low_name, T.upsample2_trilinear_backward(gu))
            else:
                n_, d_, h_, w_, c_ = low.shape
                add_grad(low_name, gu.reshape(n_, d_, 2, h_, 2, w_, 2, c_).float().sum((2, 4, 6)).to(dt))
        # data-gradient packings of the plain blocks in one launch (same record-and-replay as the forward)
        bplan = getattr(model, "_pack_plan_bwd", {}).get(ctx.pkey) if BATCH_PACK else None
        bpacks, brec = {}, {}
        pre = getattr(ctx, "bpacks_pre", None)
        if bplan and pre is not None and pre[0] == bplan:            # packed beside the forward's input import (same weights)
            bpacks = pre[1]
        elif bplan:
            bids = sorted(bplan)
            bviews = T.pack_batch([(T._as_weight(model.model[j].weight), 1, bplan[j][0], bplan[j][1]) for j in bids], dt, tensors["x"].device)
            bpacks = dict(zip(bids, bviews))

        def bpacked(j, cin_pad, width):
            brec[j] = (cin_pad, width)
            return bpacks.get(j) if bplan and bplan.get(j) == (cin_pad, width) else None

        wgrad_pending = None
        for op in reversed(ctx.ops):
            if wgrad_pending is not None:                               # the frames and the scratch are shared: one in flight
                torch.cuda.current_stream(tensors["x"].device).wait_stream(wgrad_pending)
                wgrad_pending = None
            if op[0] == "pool":
                _, src, dst, avg, pid = op
                if pid in dtap:                                         # tap at the pool id: gradient of the pooled tensor
                    if dst in grads:
                        T.import_ncdhw(dtap.pop(pid), grads[dst], accumulate=True)
                    else:
                        grads[dst] = T.import_ncdhw(dtap.pop(pid), torch.empty_like(tensors[dst]))
                if dst not in grads:
                    continue
                dp = grads.pop(dst)
                if avg:                                                 # adjoint of AvgPool3d(2): every child gets dp / 8
                    n_, d_, h_, w_, c_ = dp.shape
                    g = (dp * 0.125)[:, :, None, :, None, :, None, :].expand(n_, d_, 2, h_, 2, w_, 2, c_).reshape(
                        n_, 2 * d_, 2 * h_, 2 * w_, c_)
                    add_grad(src, g)
                elif src in grads:
                    T.pool2_max_backward(dp, tensors[src], accumulate_into=grads[src])
                else:
                    grads[src] = T.pool2_max_backward(dp, tensors[src])
                continue
            blk = op[1]
            conv, idx = blk["conv"], blk["idx"]
            x0 = tensors[blk["in0"]]
            x1 = None if blk["in1"] is None else tensors[blk["in1"]]
            n, d, h, w, c0 = x0.shape
            if blk.get("final"):
                g = dout
                rows = dtap.pop(idx, None) if idx in ctx.coords_of else None
                if g is None and rows is None:
                    continue
                if (g is None and SPARSE_OUTPUT_TAP and x1 is None and x0.is_cuda and blk["cout"] <= 16 and blk["cin"] <= 16
                        and rows.shape[1] <= 1024):
                    # the only cotangent of the output conv is 2 x 512 sampled rows: its weight and data gradient from those rows
                    # directly (amx_conv3d_backward_sampled) instead of a dense pass over a gradient volume of zeros
                    need_din = blk["in0"] != "x"
                    dw, din = T.conv_backward_sampled(rows, ctx.coords_of[idx], x0, conv.weight, blk["cin"], need_din)
                    pgrads[id(conv.weight)] = dw
                    if conv.bias is not None:
                        pgrads[id(conv.bias)] = rows.to(dt).float().sum((0, 1))
                    if need_din:
                        add_grad(blk["in0"], din)
                    elif ctx.needs_input_grad[1]:
                        raise NotImplementedError("input gradient through a sampled tap at a one-conv network")
                    continue
                fr = frame((n, d, h, w), blk["cout"])
                if g is not None:
                    T.import_ncdhw(g, T.interior(fr))
                else:
                    fr.zero_()                                            # (the whole buffer: a contiguous fill is 3x faster than the strided interior)
                if rows is not None:                                    # sampled tap at the output conv: 2 x 512 rows of gradient
                    T.scatter_rows(rows, ctx.coords_of[idx], T.interior(fr), accumulate=True)
            else:
                name = blk["name"]
                dy = grads.pop(name, None)
                bn = blk["bn"]
                for j in blk["alias_ids"]:                              # taps that alias the activated output
                    if j in dtap:
                        if dy is None:
                            dy = torch.empty_like(blk["Y"])
                            T.import_ncdhw(dtap.pop(j), dy)
                        else:
                            dy = T.import_ncdhw(dtap.pop(j), dy, accumulate=True)   # dy is owned by this backward: in place
                if dy is None and idx not in dtap:
                    continue                                            # nothing downstream of this block was used
                fr = frame((n, d, h, w), blk["cout"])
                gam = None if bn.weight is None else bn.weight.detach()
                bet = None if bn.bias is None else bn.bias.detach()
                if dy is not None and blk.get("frozen"):
                    # du = dy * act'(y) (bare activation adjoint); d gamma / d beta from the recovered pre-activation u; then the
                    # gradient of the raw convolution output is a * du and everything downstream is the ordinary conv adjoint
                    T.bn_act_backward(dy, blk["Y"], None, None, None, None, blk["act"], 0.3, framed=fr)
                    du = T.interior(fr)[..., : blk["cout"]]
                    if bn.weight is not None:
                        duf, yf = du.float(), blk["Y"][..., : blk["cout"]].float()
                        u = yf if blk["act"] != "lrelu" else torch.where(yf > 0, yf, yf / 0.3)
                        s1 = duf.sum((0, 1, 2, 3))
                        pgrads[id(bn.bias)] = s1
                        pgrads[id(bn.weight)] = ((duf * u).sum((0, 1, 2, 3)) - bn.bias.detach().float() * s1) / bn.weight.detach().float()
                    du.mul_(blk["a"].to(dt))
                elif dy is not None and isinstance(bn, nn.BatchNorm3d):
                    _, dgamma, dbeta = T.bn_act_backward(dy, blk["Y"], blk["X"], blk["mean"], blk["rstd"], gam,
                                                         blk["act"], 0.3, framed=fr, beta=bet, recompute=RECOMPUTE_ACT)
                    pgrads[id(bn.weight)], pgrads[id(bn.bias)] = dgamma, dbeta
                elif dy is not None:                                    # InstanceNorm3d: per sample
                    dgs, dbs = [], []
                    for s_ in range(n):
                        _, dg_, db_ = T.bn_act_backward(dy[s_:s_ + 1], blk["Y"][s_:s_ + 1], blk["X"][s_:s_ + 1], blk["mean"][s_],
                                                        blk["rstd"][s_], gam, blk["act"], 0.3, framed=fr[s_:s_ + 1], beta=bet,
                                                        recompute=RECOMPUTE_ACT)
                        dgs.append(dg_)
                        dbs.append(db_)
                    if bn.weight is not None:
                        pgrads[id(bn.weight)], pgrads[id(bn.bias)] = torch.stack(dgs).sum(0), torch.stack(dbs).sum(0)
                else:
                    fr.zero_()                                            # (the whole buffer: a contiguous fill is 3x faster than the strided interior)
                if idx in dtap:                                         # tap at the conv id: gradient of the PRE-norm output
                    if idx in ctx.coords_of:
                        T.scatter_rows(dtap.pop(idx), ctx.coords_of[idx], T.interior(fr), accumulate=True)
                    else:
                        T.import_ncdhw(dtap.pop(idx), T.interior(fr), accumulate=True)
            # The weight gradient and the data gradient of a block both read `fr` and nothing else of each other: the weight
            # gradient goes to a side stream (result and scratch preallocated / cached on this one) and is joined before the next
            # block touches a framed buffer (they are shared per shape).  Letting it also run beside the next block's BatchNorm
            # adjoint (second frame per shape + events) measured slower: 11.4 vs 10.7 ms per step in round 2, and again 9.04 vs 8.71 ms in
            # round 3 with the one-round weight-gradient launches (the two MFMA kernels contend; the adjoint passes lose more than the join costs).
            if OVERLAP_WGRAD and x0.is_cuda and w >= OVERLAP_WGRAD_MIN_W:
                dw = torch.empty((blk["cout"], blk["cin"], 3, 3, 3), dtype=torch.float32, device=x0.device)
                T.wgrad_scratch(x0, x1, blk["cout"])                   # make sure the cached scratch exists (allocated here)
                side = _side_stream(x0.device)
                side.wait_stream(torch.cuda.current_stream(x0.device))
                with torch.cuda.stream(side):
                    T.conv_wgrad(fr, x0, x1, blk["cin"], blk["cout"], out=dw)
                pgrads[id(conv.weight)] = dw
                wgrad_pending = side
            else:
                pgrads[id(conv.weight)] = T.conv_wgrad(fr, x0, x1, blk["cin"], blk["cout"])
            if conv.bias is not None:                                   # d bias = sum of the pre-norm gradient over the voxels
                pgrads[id(conv.bias)] = T.interior(fr).float().sum((0, 1, 2, 3))[: blk["cout"]]
            if blk["in0"] == "x":
                if ctx.needs_input_grad[1]:                             # d loss / d image: the stem's data gradient (channel 0 of
                    dx_in = T.conv_dgrad(fr, conv.weight)[..., 0].float().unsqueeze(1)   # the 16-channel padded result)
                continue
            if (x1 is not None and blk.get("cat_parts") is None and SPLIT_CONCAT_DGRAD and FUSED_FOLD_SPLIT
                    and ((blk["cout"] == 16 and c0 == 16 and x1.shape[-1] == 32) or
                         (SPLIT_CONCAT_DGRAD > 1 and c0 % 16 == 0 and x1.shape[-1] % 16 == 0))
                    and tuple(conv.weight.shape[:2]) == (blk["cout"], c0 + x1.shape[-1])):
                # the level-0 concat layer (48 -> 16): its data gradient is a 16 -> 48 convolution on the framed domain, which only the
                # generic kernel takes as one launch (415 us at 128^3 x 2 views); as a 16 -> 16 (skip channels) and a 16 -> 32
                # (upsampled channels) launch both halves run on the z-marching kernels, and the halves feed different consumers anyway
                w = conv.weight.detach()
                w_skip = w[:, :c0].contiguous()
                g_up = T.conv_dgrad_framed(fr, w[:, c0:].contiguous())
                if DIRECT_DGRAD and T.dgrad_direct_supported(fr, w_skip):
                    add_grad(blk["in0"], T.conv_dgrad_direct(fr, w_skip))
                else:
                    grads[blk["in0"]] = T.pad_fold(T.conv_dgrad_framed(fr, w_skip), grads.get(blk["in0"]))
                add_grad(blk["in1"], T.upcat_split_backward_framed(g_up, 0, x1.shape[-1])[1])
                continue
            if x1 is None and blk.get("cat_parts") is None and DIRECT_DGRAD and T.dgrad_direct_supported(fr, conv.weight):
                # the forward kernel on the interior of the framed gradient + the folded shell terms: no (n + 4)^3 domain, no fold pass
                g = T.conv_dgrad_direct(fr, conv.weight, wpk=bpacked(idx, fr.shape[-1], fr.shape[3] - 4))
                add_grad(blk["in0"], g[..., : x0.shape[-1]] if g.shape[-1] != x0.shape[-1] else g)
                continue
            g_fr = T.conv_dgrad_framed(fr, conv.weight, wpk=bpacked(idx, fr.shape[-1], fr.shape[3]))
            if (x1 is not None and blk.get("cat_parts") is None and g_fr.shape[-1] == c0 + x1.shape[-1] and c0 % 8 == 0
                    and x1.shape[-1] % 8 == 0 and FUSED_FOLD_SPLIT):
                # reflect-padding adjoint, channel split and the sum over the 8 children of every low-resolution voxel (adjoint of
                # the nearest x2 upsample) in ONE pass over the framed result; the skip part accumulates in place when the skip
                # already has a gradient
                dskip, dlow = T.upcat_split_backward_framed(g_fr, c0, x1.shape[-1], skip_into=grads.get(blk["in0"]))
                grads[blk["in0"]] = dskip
                add_grad(blk["in1"], dlow)
                continue
            dcat = T.pad_fold(g_fr)
            if blk.get("cat_parts") is not None:                        # materialised trilinear concat: split, then the adjoint
                skip_name, low_name = blk["cat_parts"]
                cs = tensors[skip_name].shape[-1]
                add_grad(skip_name, dcat[..., :cs].contiguous())
                add_grad(low_name, T.upsample2_trilinear_backward(dcat[..., cs: cs + tensors[low_name].shape[-1]]))
            elif x1 is None:
                add_grad(blk["in0"], dcat[..., : x0.shape[-1]] if dcat.shape[-1] != x0.shape[-1] else dcat)
            else:
                c1 = x1.shape[-1]
                if dcat.shape[-1] == c0 + c1 and c0 % 8 == 0 and c1 % 8 == 0:
                    # one pass: skip channels out (accumulated in place when the skip already has a gradient), the 8 children of
                    # every low-resolution voxel summed (adjoint of the nearest x2 upsample)
                    prev = grads.get(blk["in0"])
                    dskip, dlow = T.upcat_split_backward(dcat, c0, c1, skip_into=prev)
                    grads[blk["in0"]] = dskip
                    add_grad(blk["in1"], dlow)
                else:
                    add_grad(blk["in0"], dcat[..., :c0].contiguous())
                    up = dcat[..., c0:].reshape(n, d // 2, 2, h // 2, 2, w // 2, 2, c1)
                    add_grad(blk["in1"], up.float().sum((2, 4, 6)).to(dt))
        if wgrad_pending is not None:
            torch.cuda.current_stream(tensors["x"].device).wait_stream(wgrad_pending)
        if BATCH_PACK and not bplan:
            model.__dict__.setdefault("_pack_plan_bwd", {})[ctx.pkey] = brec
        return (None, dx_in, None, None) + tuple(pgrads.get(pid) for pid in ctx.param_ids)


def forward_train(model, x, layers):
    """(out, feats) like Unet.forward(input, layers) -- or out alone when ``layers`` is empty -- differentiable."""
    layers = [int(l) for l in layers]
    final_idx = max(i for i, m in enumerate(model.model) if isinstance(m, nn.Conv3d))
    want = sorted({l for l in layers if l != final_idx})
    res = _UnetTrainFn.apply(model, x, tuple(want), None, *list(model.parameters()))
    out, taps = res[0], dict(zip(want, res[1:]))
    if not layers:
        return out
    feats = [out if l == final_idx else taps[l] for l in sorted(set(layers))]
    return out, feats


def sampled_unsupported_reason(model, x, layers):
    """None when ``forward_train_sampled`` covers the request: the HIP training path itself, and every tap at a convolution that is
    followed by a norm in train mode (the pre-norm output) or at the output conv -- the ids the reference's launcher uses
    (pretraining/scripts/pretrain_anatomix.py:385: 27, 31, 38, 45, 52, 65)."""
    r = unsupported_reason(model, x, list(layers))
    if r is not None:
        return r
    kinds = _module_kinds(model)
    mods = list(model.model)
    for l in layers:
        if kinds[l] != "conv":
            return "sampled taps are implemented at conv ids (pre-norm outputs) and the output conv"
        if l + 1 < len(mods) and kinds[l + 1] == "norm" and isinstance(mods[l + 1], nn.BatchNorm3d) and not mods[l + 1].training:
            return "sampled taps at a frozen-statistics block"
    return None


def forward_train_sampled(model, x, layers, sampler):
    """The differentiable train-mode forward with SAMPLED taps: returns ``(out, rows, coords, dims)`` -- per tapped module id (ascending) the
    fp32 rows [N, P, C] at the coordinates ``sampler(module id, (d, h, w))`` drew when the forward reached that tensor, those
    coordinates, and the tensor's spatial size.  Same values as gathering from ``forward_train``'s dense taps; the dense fp32 copies, their
    zero-filled gradients and the import passes are never made."""
    layers = sorted({int(l) for l in layers})
    dims, coords = {}, {}

    def recording(i, shape):
        dims[i] = tuple(shape)
        coords[i] = sampler(i, shape)
        return coords[i]

    if getattr(sampler, "on_start", None) is not None:
        recording.on_start = sampler.on_start
    res = _UnetTrainFn.apply(model, x, tuple(layers), recording, *list(model.parameters()))
    return res[0], list(res[1:]), [coords[l] for l in layers], [dims[l] for l in layers]
