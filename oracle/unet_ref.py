"""Oracle: functional CPU restatement of ``anatomix.model.network.Unet.forward``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against the imported
reference by ``oracle/make_golden.py`` (run in the build container, where
/root/reference exists) and against the committed fixtures in ``tests/golden``.

Reference lines restated here (paths relative to /root/reference):
  * layer list / channel plan ............ anatomix/model/network.py:309-465
  * standard forward (skip-first concat) . anatomix/model/network.py:530-548
  * ``layers`` / ``encode_only`` branch .. anatomix/model/network.py:475-529
  * norm / activation factories .......... anatomix/model/network.py:127-204
  * variant kwargs ....................... anatomix/model/load_from_hf.py:11-36

All arithmetic is torch.nn.functional on CPU tensors (the same ATen kernels the
reference's nn.Modules dispatch to).  ``dtype`` may be float32 or float64.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

VARIANTS = {
    # anatomix/model/load_from_hf.py:12-17
    "anatomix": dict(dimension=3, input_nc=1, output_nc=16, num_downs=4, ngf=16),
    # anatomix/model/load_from_hf.py:18-24
    "anatomix-dev": dict(dimension=3, input_nc=1, output_nc=32, num_downs=5, ngf=32,
                         norm="instance", pooling="Avg", interp="trilinear", norm_eps=1e-2),
}


@dataclass
class Plan:
    """Module-index plan of the nn.Sequential the reference builds."""
    kinds: List[str] = field(default_factory=list)        # 'conv','norm','act','pool','up','final_act'
    conv_io: Dict[int, Tuple[int, int]] = field(default_factory=dict)   # idx -> (cin, cout)
    norm_c: Dict[int, int] = field(default_factory=dict)
    encoder_idx: List[int] = field(default_factory=list)
    decoder_idx: List[int] = field(default_factory=list)
    use_bias: bool = False


def build_plan(input_nc, output_nc, num_downs, ngf=24, norm="batch", final_act="none",
               activation="relu", doubleconv=True, use_skip_connection=True, **_unused) -> Plan:
    """Replays the constructor's list building (network.py:309-465) without nn.Modules."""
    p = Plan(use_bias=(norm == "instance"))          # network.py:292
    has_norm = norm != "none"
    has_act = activation != "none"

    def cna(cin, cout):
        p.conv_io[len(p.kinds)] = (cin, cout); p.kinds.append("conv")
        if has_norm:
            p.norm_c[len(p.kinds)] = cout; p.kinds.append("norm")
        if has_act:
            p.kinds.append("act")

    cna(input_nc, ngf)                                # network.py:309-326
    in_ngf = ngf
    for i in range(num_downs):                        # network.py:334-369
        mult = 1 if i == 0 else 2
        cna(in_ngf, in_ngf * mult)
        if doubleconv:
            cna(in_ngf * mult, in_ngf * mult)
        p.encoder_idx.append(len(p.kinds) - 1)
        p.kinds.append("pool")
        in_ngf *= mult
    cna(in_ngf, in_ngf * 2)                           # network.py:372-400
    if doubleconv:
        cna(in_ngf * 2, in_ngf * 2)
    mult = 2 ** num_downs
    for i in range(num_downs):                        # network.py:403-445
        p.decoder_idx.append(len(p.kinds))
        p.kinds.append("up")
        m = mult + mult // 2 if use_skip_connection else mult
        cna(ngf * m, ngf * (mult // 2))
        if doubleconv:
            cna(ngf * (mult // 2), ngf * (mult // 2))
        mult //= 2
    p.conv_io[len(p.kinds)] = (ngf * mult, output_nc); p.kinds.append("conv")   # network.py:452-461
    if final_act != "none":
        p.kinds.append("final_act")
    return p


def synthetic_state_dict(kwargs: dict, seed: int, gain: float = 1.0) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic non-trivial parameters.  Regenerated from the seed, never stored.

    conv w ~ N(0, gain/sqrt(27*Cin)) (gain=1: SURVEY spec; gain=sqrt(2): He-scaled, keeps the signal
    alive through all 20 layers and is the harder, less damped test); conv bias ~ N(0,0.1) when present;
    BN gamma ~ U(0.5,1.5), beta ~ N(0,0.1), running_mean ~ N(0,0.1), running_var ~ U(0.5,1.5).
    Keys / shapes / dtypes are exactly those of the reference module's state_dict.
    """
    rs = np.random.RandomState(seed)
    p = build_plan(**{k: v for k, v in kwargs.items() if k != "dimension"})
    norm = kwargs.get("norm", "batch")
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for i, kind in enumerate(p.kinds):
        if kind == "conv":
            cin, cout = p.conv_io[i]
            w = rs.randn(cout, cin, 3, 3, 3) * (gain / np.sqrt(27.0 * cin))
            sd[f"model.{i}.weight"] = torch.from_numpy(w.astype(np.float32))
            if p.use_bias:
                sd[f"model.{i}.bias"] = torch.from_numpy((rs.randn(cout) * 0.1).astype(np.float32))
        elif kind == "norm" and norm == "batch":
            c = p.norm_c[i]
            sd[f"model.{i}.weight"] = torch.from_numpy(rs.uniform(0.5, 1.5, c).astype(np.float32))
            sd[f"model.{i}.bias"] = torch.from_numpy((rs.randn(c) * 0.1).astype(np.float32))
            sd[f"model.{i}.running_mean"] = torch.from_numpy((rs.randn(c) * 0.1).astype(np.float32))
            sd[f"model.{i}.running_var"] = torch.from_numpy(rs.uniform(0.5, 1.5, c).astype(np.float32))
            sd[f"model.{i}.num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)
        elif kind == "norm" and norm == "instance_affine":
            c = p.norm_c[i]
            sd[f"model.{i}.weight"] = torch.from_numpy(rs.uniform(0.5, 1.5, c).astype(np.float32))
            sd[f"model.{i}.bias"] = torch.from_numpy((rs.randn(c) * 0.1).astype(np.float32))
    return sd


def synthetic_input(seed: int, n: int, size: Sequence[int]) -> torch.Tensor:
    """Input volume: RandomState(seed).rand(n,1,D,H,W) float32 in [0,1)."""
    d, h, w = size
    return torch.from_numpy(np.random.RandomState(seed).rand(n, 1, d, h, w).astype(np.float32))


# ---------------------------------------------------------------------------------------------
# ops (each cites the nn.Module the reference instantiates)
# ---------------------------------------------------------------------------------------------
def conv3_reflect(x, w, b=None):
    """nn.Conv3d(k=3, stride=1, padding='same', padding_mode='reflect') -- network.py:310-318.
    ATen lowers this to F.pad(..., mode='reflect') + valid cross-correlation."""
    return F.conv3d(F.pad(x, (1, 1, 1, 1, 1, 1), mode="reflect"), w, b)


def norm_apply(x, sd, i, norm, eps):
    """get_norm_layer -- network.py:127-168.  Batch norm in EVAL mode (running stats)."""
    if norm == "batch":
        g, b = sd[f"model.{i}.weight"], sd[f"model.{i}.bias"]
        m, v = sd[f"model.{i}.running_mean"], sd[f"model.{i}.running_var"]
        return F.batch_norm(x, m.to(x.dtype), v.to(x.dtype), g.to(x.dtype), b.to(x.dtype), False, 0.1, eps)
    if norm == "instance":
        return F.instance_norm(x, eps=eps)
    if norm == "instance_affine":
        return F.instance_norm(x, weight=sd[f"model.{i}.weight"].to(x.dtype),
                               bias=sd[f"model.{i}.bias"].to(x.dtype), eps=eps)
    raise ValueError(norm)


def act_apply(x, activation):
    """get_actvn_layer -- network.py:171-204 (relu, lrelu slope 0.3).  relu / lrelu / selu are built with
    inplace=True there (188-196): they OVERWRITE their input, so a feature tap taken at the preceding norm (or, with
    norm='none', conv) module aliases the activated values.  The in-place functional forms keep that behaviour."""
    if activation == "relu":
        return F.relu_(x)
    if activation == "lrelu":
        return F.leaky_relu_(x, 0.3)
    if activation == "tanh":
        return torch.tanh(x)
    if activation == "elu":
        return F.elu(x)
    if activation == "selu":
        return F.selu(x, inplace=True)
    raise ValueError(activation)


def forward(x: torch.Tensor, sd: dict, kwargs: dict, layers: Sequence[int] = (), encode_only: bool = False,
            dtype=torch.float32):
    """Restates Unet.forward (network.py:467-548), both branches, on CPU."""
    kw = dict(ngf=24, norm="batch", final_act="none", activation="relu", pad_type="reflect",
              doubleconv=True, residual_connection=False, pooling="Max", interp="nearest",
              use_skip_connection=True, norm_eps=1e-5)
    kw.update(kwargs)
    assert kw["pad_type"] == "reflect" and not kw["residual_connection"]
    p = build_plan(**{k: v for k, v in kw.items() if k != "dimension"})
    layers = list(layers)
    feat = x.to(dtype)
    feats, skips = [], []
    for i, kind in enumerate(p.kinds):
        if kind == "conv":
            w = sd[f"model.{i}.weight"].to(dtype)
            b = sd.get(f"model.{i}.bias")
            feat = conv3_reflect(feat, w, None if b is None else b.to(dtype))
        elif kind == "norm":
            feat = norm_apply(feat, sd, i, kw["norm"], kw["norm_eps"])
        elif kind == "act":
            feat = act_apply(feat, kw["activation"])
        elif kind == "final_act":
            feat = act_apply(feat, kw["final_act"])
        elif kind == "pool":                      # network.py:297,368
            feat = F.max_pool3d(feat, 2) if kw["pooling"] == "Max" else F.avg_pool3d(feat, 2)
        elif kind == "up":                        # network.py:407
            feat = F.interpolate(feat, scale_factor=2, mode=kw["interp"])
        if kw["use_skip_connection"]:
            if i in p.encoder_idx:
                skips.append(feat)
            if i in p.decoder_idx:                # skip channels FIRST -- network.py:502,545
                feat = torch.cat((skips.pop(), feat), dim=1)
        if layers:
            if i in layers:
                feats.append(feat)
            if encode_only and i == layers[-1]:
                return feats
    return (feat, feats) if layers else feat


# ---------------------------------------------------------------------------------------------
# Storage-precision emulation: the same graph with the HIP path's rounding points
# ---------------------------------------------------------------------------------------------
def fold_conv_params(sd, kwargs, i, plan: Plan, dtype=torch.float64):
    """(weight * gain, shift) of conv ``i`` with the eval-mode BatchNorm that follows it folded in."""
    w = sd[f"model.{i}.weight"].to(dtype)
    cb = sd.get(f"model.{i}.bias")
    norm = kwargs.get("norm", "batch")
    has_norm = i + 1 < len(plan.kinds) and plan.kinds[i + 1] == "norm"
    if has_norm and norm == "batch":
        g, b = sd[f"model.{i+1}.weight"].to(dtype), sd[f"model.{i+1}.bias"].to(dtype)
        m, v = sd[f"model.{i+1}.running_mean"].to(dtype), sd[f"model.{i+1}.running_var"].to(dtype)
        s = g / torch.sqrt(v + kwargs.get("norm_eps", 1e-5))
        t = b - m * s
    else:
        s = torch.ones(w.shape[0], dtype=dtype)
        t = torch.zeros(w.shape[0], dtype=dtype)
    if cb is not None:
        t = t + cb.to(dtype) * s
    return w * s[:, None, None, None, None], t


def forward_lowp(x: torch.Tensor, sd: dict, kwargs: dict, lowp=torch.float16):
    """Emulates the HIP path's numerics on CPU (all norm / interp / pooling modes): folded
    weights and every stored activation are rounded to ``lowp`` (fp16 or bf16), products are
    accumulated in fp32, shift + activation are applied in fp32 before the store rounding.  The
    final conv output stays fp32.  Used to separate kernel bugs (must match this to ~1e-4) from
    the storage-precision error (distance of this from ``forward``)."""
    kw = dict(ngf=24, norm="batch", final_act="none", activation="relu", pooling="Max", interp="nearest",
              use_skip_connection=True, norm_eps=1e-5, doubleconv=True)
    kw.update(kwargs)
    inorm = kw["norm"] in ("instance", "instance_affine")
    p = build_plan(**{k: v for k, v in kw.items() if k != "dimension"})
    q = lambda t: t.to(lowp).to(torch.float32)
    feat = q(x.float())
    skips = []
    i, n = 0, len(p.kinds)
    while i < n:
        kind = p.kinds[i]
        last = i
        if kind == "conv":
            w, t = fold_conv_params(sd, kw, i, p)
            feat = conv3_reflect(feat, q(w.float()), t.float())
            j = i + 1
            if j < n and p.kinds[j] == "norm":
                if inorm:      # the HIP path stores the raw conv output, then normalises it in place (fp32 math)
                    feat = norm_apply(q(feat), sd, j, kw["norm"], kw["norm_eps"])
                j += 1
            if j < n and p.kinds[j] == "act":
                feat = act_apply(feat, kw["activation"]); j += 1
            is_final = i == max(p.conv_io)
            if is_final:
                if j < n and p.kinds[j] == "final_act":
                    feat = act_apply(feat, kw["final_act"]); j += 1
            else:
                feat = q(feat)
            last = j - 1
            i = j
        elif kind == "pool":
            feat = F.max_pool3d(feat, 2) if kw["pooling"] == "Max" else q(F.avg_pool3d(feat, 2))
            i += 1
        elif kind == "up":
            feat = F.interpolate(feat, scale_factor=2, mode="nearest") if kw["interp"] == "nearest" else \
                q(F.interpolate(feat, scale_factor=2, mode=kw["interp"]))
            i += 1
        else:
            i += 1
        if kw["use_skip_connection"]:
            if last in p.encoder_idx:
                skips.append(feat)
            if last in p.decoder_idx:
                feat = torch.cat((skips.pop(), feat), dim=1)
    return feat
