"""TEST INFRASTRUCTURE (never imported by the product): the training graph of the 6 M UNet with the HIP training path's ROUNDING
POINTS, fp32 arithmetic in between -- what `unet_ref.forward_lowp` is for the inference forward.  It separates "the backward
kernels compute the wrong thing" (must match this closely) from "16-bit storage moves a 20-layer train-mode BatchNorm network"
(the distance of this from the fp32 modules, tens of percent per parameter: tests/test_train_step_gpu.py TRAIN_BOUNDS).

Follows anatomix_amd/model/train.py (`_UnetTrainFn`), which itself follows /root/reference/anatomix/model/network.py:467-548
(forward) under /root/reference/pretraining/models/supcl_model.py:603-661 (backward through the taps):

  forward   x -> 16 bit.  Per conv -> BatchNorm(train) -> act block: X = r16(conv(reflect_pad(in), r16(W))) is STORED; batch
            statistics from the stored X (fp32); Y = r16(act(a X + b)), a = gamma rstd, b = beta - mean a, is STORED.  Max-pool,
            nearest upsample and the skip concat move stored values unchanged.  The output conv stays fp32.
  backward  every gradient tensor the path keeps is 16 bit: the cotangent of the output / of a tap as imported; dX (the norm
            adjoint's result, in the framed buffer) -- a tap gradient at the conv id is added to it there; the data gradient on
            the PADDED domain (the framed conv result) and again after the reflect fold / the sum over the eight children of an
            upsampled voxel; the sum of two contributions to one tensor (skip connections).  Weight gradients and the norm's
            d gamma / d beta are fp32 sums of products of stored 16-bit operands.

Not emulated (noise, not structure): the merged-tap layers round SUMS of taps once instead of each tap (forward weights of the
upsampled channels differ by <= 1 ulp); fp32 summation order.
"""
import torch
import torch.nn.functional as F

from . import unet_ref as R


def _r(t, dt):
    return t.to(dt).to(torch.float32)


class _RoundBoth(torch.autograd.Function):          # a stored activation: value and gradient live in 16 bit
    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return _r(x, dt)

    @staticmethod
    def backward(ctx, g):
        return _r(g, ctx.dt), None


class _RoundBwd(torch.autograd.Function):           # a use of a stored tensor: its gradient contribution is stored in 16 bit
    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g, ctx.dt), None


class _RoundFwd(torch.autograd.Function):           # packed weights: 16-bit operands, fp32 gradient
    @staticmethod
    def forward(ctx, w, dt):
        return _r(w, dt)

    @staticmethod
    def backward(ctx, g):
        return g, None


def _force(t, forced, idx):
    """Teacher forcing: the VALUE of a stored tensor is replaced by the one the HIP path stored (forced[idx], fp32 NCDHW), its gradient
    keeps flowing into the emulated graph.  A random-weight train-mode BatchNorm network doubles any 1-ulp storage flip per layer
    (7e-6 at the stem, 1.4e-2 at module 62 in f16, tools/train_lowp_debug.py): without forcing, the comparison of two correct
    implementations reads like a bug at depth; with it every layer's adjoint is checked on the SAME stored operands."""
    if forced is None or idx not in forced:
        return t
    return t + (forced[idx].to(t.dtype) - t).detach()


def forward_train_lowp(x, params, kwargs, layers=(), lowp=torch.bfloat16, forced=None):
    """x [N, 1, D, H, W] fp32; params: {state-dict key: fp32 leaf tensor requiring grad} (conv weights, BatchNorm weight / bias);
    returns (out fp32, [taps in ascending module order]) -- taps at conv ids are the stored PRE-norm tensors, at norm / act ids the
    stored activated tensors, like model/train.py.  forced: {module id: tensor} values of stored tensors to substitute (conv id: X,
    last id of its norm / act group: Y); see _force."""
    kw = dict(ngf=24, norm="batch", final_act="none", activation="relu", pooling="Max", interp="nearest",
              use_skip_connection=True, norm_eps=1e-5, doubleconv=True)
    kw.update(kwargs)
    assert kw["norm"] == "batch" and kw["activation"] == "relu" and kw["pooling"] == "Max" and kw["interp"] == "nearest"
    p = R.build_plan(**{k: v for k, v in kw.items() if k != "dimension"})
    use = lambda t: _RoundBwd.apply(t, lowp)
    cur = _r(x.float(), lowp)
    skips, taps = [], {}
    fresh_cat = False
    i, n = 0, len(p.kinds)
    final = max(p.conv_io)
    while i < n:
        kind = p.kinds[i]
        last = i
        if kind == "conv":
            w = _RoundFwd.apply(params[f"model.{i}.weight"], lowp)
            # gradient on the padded domain: 16 bit (the framed conv result); after the reflect fold: 16 bit again for a plain input --
            # a concat input is folded, split and (upsampled part) summed over the eight children in ONE fp32 pass, rounded per part
            inp = cur if fresh_cat else use(cur)
            fresh_cat = False
            z = F.conv3d(use(F.pad(inp, (1,) * 6, mode="reflect")), w)
            if i == final:
                out = use(z)                                    # fp32 output; its cotangent is imported to 16 bit
                cur = out
                if i in layers:
                    taps[i] = out
                i += 1
                continue
            X = _force(_RoundBoth.apply(z, lowp), forced, i)
            if i in layers:
                taps[i] = X
            j = i + 1
            assert p.kinds[j] == "norm"
            xn = use(X)
            mean = xn.double().mean((0, 2, 3, 4)).float()
            var = (xn.double() - mean.double().view(1, -1, 1, 1, 1)).square().mean((0, 2, 3, 4)).float()
            rstd = (var + kw["norm_eps"]).rsqrt()
            a = params[f"model.{j}.weight"] * rstd
            b = params[f"model.{j}.bias"] - mean * a
            y = xn * a.view(1, -1, 1, 1, 1) + b.view(1, -1, 1, 1, 1)
            j += 1
            if j < n and p.kinds[j] == "act":
                y = torch.relu(y)
                j += 1
            Y = _force(_RoundBoth.apply(y, lowp), forced, j - 1)
            for t in range(i + 1, j):
                if t in layers:
                    taps[t] = Y
            cur = Y
            last = j - 1
            i = j
        elif kind == "pool":
            cur = F.max_pool3d(use(cur), 2)
            if i in layers:
                taps[i] = cur
            i += 1
        elif kind == "up":
            cur = F.interpolate(use(cur), scale_factor=2, mode="nearest")
            i += 1
        else:
            i += 1
        if last in p.encoder_idx:
            skips.append(cur)
        if last in p.decoder_idx:
            cur = torch.cat((use(skips.pop()), cur), dim=1)
            fresh_cat = True
    return out, [taps[l] for l in sorted(taps)]


def parameter_gradients(x, sd, kwargs, layers, cotangents, out_weight=0.1, lowp=torch.bfloat16, forced=None):
    """Gradients of  out_weight * mean(out^2) + sum_i <tap_i, cotangent_i>  with respect to every conv weight and BatchNorm weight /
    bias of the state dict `sd` (fp32 tensors), through forward_train_lowp.  Returns ({key: gradient}, out, taps)."""
    params = {k: v.detach().clone().float().requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and ("running" not in k)}
    out, taps = forward_train_lowp(x, params, kwargs, layers, lowp, forced)
    loss = out_weight * out.square().mean()
    for t, c in zip(taps, cotangents):
        loss = loss + (t * c).sum()
    loss.backward()
    return {k: v.grad for k, v in params.items() if v.grad is not None}, out.detach(), [t.detach() for t in taps]
