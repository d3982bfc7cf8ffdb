"""Oracle: functional CPU restatement of the 3D ViT variant ``anatomix-dev-vit`` (PrimusV2-S, BASELINE configs[4]).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **PARITY UNPINNED.**

The reference's own code for this model is a thin wrapper (anatomix/model/vit3d/architectures.py:89-165, 231-260,
deep_tokenizer.py:12-149, registry entry load_from_hf.py:25-35) over the third-party package
``dynamic-network-architectures==0.4.4`` (requirements.txt:17) and, through it, ``timm``'s EVA blocks.  Neither package is
vendored under /root/reference nor installed in this image, the reference ships no test or golden vector for the model, and
its checkpoint is not in the tree -- so the arithmetic below restates the PUBLISHED algorithms (Primus: Wald et al.,
arXiv:2503.01835; EVA-02 blocks: Fang et al. 2023 as implemented in timm's ``eva.py``: sub-LN SwiGLU MLP, rotary position
embedding concatenated as [sin | cos], LayerScale) plus exactly what the reference's wrapper adds on top:
  * per-head LayerNorm of queries and keys before the rotary embedding (architectures.py:108-115, ``qk_norm``);
  * register tokens prepended to the patch tokens and dropped before decoding, re-initialised with ``register_init_std``
    (architectures.py:117-120);
  * output normalisation ``ChannelDemean`` (architectures.py:28-33, ``out_norm="demean"``);
  * InstanceNorm epsilon ``in_eps`` in every tokenizer norm (architectures.py:252-255, deep_tokenizer.py:66-68);
  * the UNet-compatible call contract ``forward(x, layers, encode_only)`` (architectures.py:122-165).
What cannot be checked here (layer names of the upstream state_dict, the tokenizer's exact stage widths, the decoder's
channel schedule, the rotary frequency bands) is fixed by ``vit_plan`` below and used identically by the product module
(anatomix_amd/model/vit3d) -- the GPU tests compare the HIP path against THIS restatement, which says nothing about the
upstream package.  Every such choice is marked "(assumed)".
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# load_from_hf.py:25-35
VIT_VARIANTS = {
    "anatomix-dev-vit": dict(input_channels=1, num_classes=32, embed_dim=396, eva_depth=12, eva_numheads=6,
                             patch_embed_size=(8, 8, 8), input_shape=(128, 128, 128), num_register_tokens=8, init_values=0.1,
                             scale_attn_inner=True, qk_norm=True, out_norm="demean", out_norm_eps=1e-2, register_init_std=0.02,
                             in_eps=1e-2),
}


def vit_plan(kw):
    """Shapes of every parameter group (shared with the product module so both build the same network)."""
    e, heads = kw["embed_dim"], kw["eva_numheads"]
    base = 32                                                   # tokenizer base_features (deep_tokenizer.py:47)
    stages = [base, 2 * base, 4 * base]                         # three stride-2 residual stages (assumed: width doubles)
    hidden = int(e * 4 * 2 / 3)                                 # SwiGLU hidden width, mlp_ratio = 4 * 2 / 3 (assumed)
    nst = int(round(math.log2(max(kw["patch_embed_size"]))))    # decoder: one x2 transposed conv per factor of two
    red = (e / (2 * kw["num_classes"])) ** (1.0 / nst)
    r8 = lambda v: int(max(8, round((v + 1e-6) / 8) * 8))
    dec = [e] + [r8(e / red ** (k + 1)) for k in range(nst)]
    dec[-1] = kw["num_classes"]
    return dict(base=base, stages=stages, hidden=hidden, dec=dec, head_dim=e // heads,
                grid=tuple(s // p for s, p in zip(kw["input_shape"], kw["patch_embed_size"])))


def synthetic_state_dict(kw, seed):
    """Deterministic parameters, regenerated from the seed (keys are those of anatomix_amd.model.vit3d.PrimusV2)."""
    rs = np.random.RandomState(seed)
    pl = vit_plan(kw)
    e, hd, hid = kw["embed_dim"], pl["head_dim"], pl["hidden"]
    sd = OrderedDict()
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    conv = lambda co, ci, k: t(rs.randn(co, ci, k, k, k) / math.sqrt(ci * k ** 3))
    vec = lambda c, s=0.1: t(rs.randn(c) * s)
    gain = lambda c: t(rs.uniform(0.5, 1.5, c))
    lin = lambda o, i: t(rs.randn(o, i) / math.sqrt(i))

    def norm(prefix, c):
        sd[prefix + ".weight"], sd[prefix + ".bias"] = gain(c), vec(c)

    sd["down_projection.stem.conv.weight"], sd["down_projection.stem.conv.bias"] = conv(pl["base"], kw["input_channels"], 3), vec(pl["base"])
    norm("down_projection.stem.norm", pl["base"])
    cin = pl["base"]
    for k, c in enumerate(pl["stages"]):
        p = f"down_projection.stages.{k}"
        sd[p + ".conv1.weight"], sd[p + ".conv1.bias"] = conv(c, cin, 3), vec(c)
        norm(p + ".norm1", c)
        sd[p + ".conv2.weight"], sd[p + ".conv2.bias"] = conv(c, c, 3), vec(c)
        norm(p + ".norm2", c)
        sd[p + ".skip.weight"] = conv(c, cin, 1)
        norm(p + ".skip_norm", c)
        cin = c
    sd["down_projection.proj.weight"], sd["down_projection.proj.bias"] = conv(e, cin, 1), vec(e)
    sd["register_tokens"] = t(rs.randn(1, kw["num_register_tokens"], e) * kw["register_init_std"])
    sd["eva.pos_embed"] = t(rs.randn(1, int(np.prod(pl["grid"])), e) * 0.02)
    for b in range(kw["eva_depth"]):
        p = f"eva.blocks.{b}"
        norm(p + ".norm1", e)
        for nm in ("q_proj", "k_proj", "v_proj", "proj"):
            sd[f"{p}.attn.{nm}.weight"] = lin(e, e)
            if nm != "k_proj":                                  # EVA: no key bias
                sd[f"{p}.attn.{nm}.bias"] = vec(e)
        norm(p + ".attn.q_norm", hd)
        norm(p + ".attn.k_norm", hd)
        norm(p + ".attn.norm", e)                               # scale_attn_inner
        sd[p + ".gamma_1"] = t(np.full(e, kw["init_values"]) * rs.uniform(0.5, 1.5, e))
        norm(p + ".norm2", e)
        sd[p + ".mlp.fc1_g.weight"], sd[p + ".mlp.fc1_g.bias"] = lin(hid, e), vec(hid)
        sd[p + ".mlp.fc1_x.weight"], sd[p + ".mlp.fc1_x.bias"] = lin(hid, e), vec(hid)
        norm(p + ".mlp.norm", hid)
        sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"] = lin(e, hid), vec(e)
        sd[p + ".gamma_2"] = t(np.full(e, kw["init_values"]) * rs.uniform(0.5, 1.5, e))
    norm("eva.norm", e)
    dec = pl["dec"]
    for k in range(len(dec) - 1):
        p = f"up_projection.decode.{k}"
        last = k == len(dec) - 2
        w = t(rs.randn(dec[k], dec[k + 1], 2, 2, 2) / math.sqrt(dec[k]))
        sd[(p if last else p + ".0") + ".weight"], sd[(p if last else p + ".0") + ".bias"] = w, vec(dec[k + 1])
        if not last:
            norm(p + ".1", dec[k + 1])
    return sd


def synthetic_input(seed, n, size=(128, 128, 128)):
    return torch.from_numpy(np.random.RandomState(seed).rand(n, 1, *size).astype(np.float32))


# ---------------------------------------------------------------------------------------------------------------------
def rope_table(grid, head_dim, temperature=10000.0, dtype=torch.float32):
    """[N, 2 * head_dim]: per token [sin (head_dim) | cos (head_dim)], every frequency repeated for its (even, odd) channel
    pair (timm ``build_rotary_pos_embed`` / ``RotaryEmbeddingCat``, in_pixels=False; 3 axes x head_dim / 6 bands (assumed))."""
    nb = head_dim // (2 * len(grid))
    bands = 1.0 / (temperature ** (torch.arange(nb, dtype=torch.float64) / nb))
    axes = torch.meshgrid(*[torch.arange(s, dtype=torch.float64) for s in grid], indexing="ij")
    pos = torch.stack(axes, dim=-1).reshape(-1, len(grid), 1) * bands          # [N, 3, nb]
    ang = pos.reshape(pos.shape[0], -1)
    sin, cos = ang.sin().repeat_interleave(2, -1), ang.cos().repeat_interleave(2, -1)
    return torch.cat((sin, cos), -1).to(dtype)


def apply_rope(x, table):
    """timm ``apply_rot_embed_cat``: x * cos + rot(x) * sin, rot(x) = (-x_odd, x_even) interleaved."""
    hd = x.shape[-1]
    sin, cos = table[:, :hd], table[:, hd:]
    rot = torch.stack((-x[..., 1::2], x[..., ::2]), -1).reshape(x.shape)
    return x * cos + rot * sin


def attention(x, sd, p, heads, table, n_prefix, qk_norm=True, scale_attn_inner=True):
    """EvaAttention with separate q / k / v projections (no key bias), per-head QK LayerNorm (architectures.py:108-115), rotary
    embedding on the patch tokens only, softmax(q k^T / sqrt(d)) v, inner LayerNorm, output projection."""
    B, N, E = x.shape
    hd = E // heads
    q = F.linear(x, sd[p + ".q_proj.weight"], sd[p + ".q_proj.bias"]).reshape(B, N, heads, hd).transpose(1, 2)
    k = F.linear(x, sd[p + ".k_proj.weight"]).reshape(B, N, heads, hd).transpose(1, 2)
    v = F.linear(x, sd[p + ".v_proj.weight"], sd[p + ".v_proj.bias"]).reshape(B, N, heads, hd).transpose(1, 2)
    if qk_norm:
        q = F.layer_norm(q, (hd,), sd[p + ".q_norm.weight"], sd[p + ".q_norm.bias"], 1e-5)
        k = F.layer_norm(k, (hd,), sd[p + ".k_norm.weight"], sd[p + ".k_norm.bias"], 1e-5)
    q = torch.cat((q[:, :, :n_prefix], apply_rope(q[:, :, n_prefix:], table)), 2)
    k = torch.cat((k[:, :, :n_prefix], apply_rope(k[:, :, n_prefix:], table)), 2)
    att = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v
    y = att.transpose(1, 2).reshape(B, N, E)
    if scale_attn_inner:
        y = F.layer_norm(y, (E,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)
    return F.linear(y, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def tokenizer(x, sd, kw):
    """PatchEmbed_deeper as configured by deep_tokenizer.py:44-68: stem conv-IN-LeakyReLU, three stride-2 BasicBlockD residual
    stages (conv-IN-LeakyReLU-conv-IN + [AvgPool(2) -> 1x1 conv -> IN] skip, LeakyReLU(0.01)), 1x1x1 projection to embed_dim."""
    eps = kw["in_eps"]
    inorm = lambda t, p: F.instance_norm(t, weight=sd[p + ".weight"], bias=sd[p + ".bias"], eps=eps)
    x = F.leaky_relu(inorm(F.conv3d(x, sd["down_projection.stem.conv.weight"], sd["down_projection.stem.conv.bias"], padding=1),
                           "down_projection.stem.norm"), 0.01)
    for k in range(3):
        p = f"down_projection.stages.{k}"
        y = F.leaky_relu(inorm(F.conv3d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], stride=2, padding=1), p + ".norm1"), 0.01)
        y = inorm(F.conv3d(y, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1), p + ".norm2")
        s = inorm(F.conv3d(F.avg_pool3d(x, 2), sd[p + ".skip.weight"]), p + ".skip_norm")
        x = F.leaky_relu(y + s, 0.01)
    return F.conv3d(x, sd["down_projection.proj.weight"], sd["down_projection.proj.bias"])


def decoder(x, sd, kw):
    """PatchDecode: one ConvTranspose3d(k=2, s=2) per factor of two, channel-wise LayerNorm (LayerNormNd, eps 1e-6) + GELU
    between them, none after the last."""
    n = len(vit_plan(kw)["dec"]) - 1
    for k in range(n):
        p = f"up_projection.decode.{k}"
        last = k == n - 1
        x = F.conv_transpose3d(x, sd[(p if last else p + ".0") + ".weight"], sd[(p if last else p + ".0") + ".bias"], stride=2)
        if not last:
            u = x.mean(1, keepdim=True)
            s = (x - u).pow(2).mean(1, keepdim=True)
            x = (x - u) / torch.sqrt(s + 1e-6)
            x = x * sd[p + ".1.weight"][None, :, None, None, None] + sd[p + ".1.bias"][None, :, None, None, None]
            x = F.gelu(x)
    return x


def forward(x, sd, kw, layers=None, encode_only=False, dtype=torch.float32):
    """PrimusV2.forward through the reference wrapper's call contract (architectures.py:122-165)."""
    sd = {k: v.to(dtype) for k, v in sd.items()}
    pl = vit_plan(kw)
    heads, nreg = kw["eva_numheads"], kw["num_register_tokens"]
    feat = tokenizer(x.to(dtype), sd, kw)                                    # [B, E, w, h, d]
    B, E = feat.shape[:2]
    grid = tuple(feat.shape[2:])
    tok = feat.flatten(2).transpose(1, 2)                                   # tokens in raster order of the embedding grid (assumed)
    tok = tok + sd["eva.pos_embed"]
    tok = torch.cat((sd["register_tokens"].expand(B, -1, -1), tok), 1)
    table = rope_table(grid, pl["head_dim"], dtype=dtype)
    for b in range(kw["eva_depth"]):
        p = f"eva.blocks.{b}"
        h = F.layer_norm(tok, (E,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-6)
        tok = tok + sd[p + ".gamma_1"] * attention(h, sd, p + ".attn", heads, table, nreg, kw["qk_norm"], kw["scale_attn_inner"])
        h = F.layer_norm(tok, (E,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-6)
        g = F.silu(F.linear(h, sd[p + ".mlp.fc1_g.weight"], sd[p + ".mlp.fc1_g.bias"])) * \
            F.linear(h, sd[p + ".mlp.fc1_x.weight"], sd[p + ".mlp.fc1_x.bias"])
        g = F.layer_norm(g, (g.shape[-1],), sd[p + ".mlp.norm.weight"], sd[p + ".mlp.norm.bias"], 1e-6)
        tok = tok + sd[p + ".gamma_2"] * F.linear(g, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
    tok = F.layer_norm(tok, (E,), sd["eva.norm.weight"], sd["eva.norm.bias"], 1e-6)
    tok = tok[:, nreg:]
    vol = tok.transpose(1, 2).reshape(B, E, *grid)
    out = decoder(vol, sd, kw)
    if kw["out_norm"] == "demean":                                           # ChannelDemean, architectures.py:28-33
        out = out - out.mean(dim=(2, 3, 4), keepdim=True)
    if layers:
        return [out] if encode_only else (out, [out])
    return out
