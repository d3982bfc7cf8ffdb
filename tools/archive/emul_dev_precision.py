"""CPU emulation of candidate reduced-work arithmetic for the compliant `anatomix-dev` forward (round 4).

Question: which arithmetic with FEWER executed MFMAs than the 3-product hi/lo split still holds
the 1e-3 tolerance on the InstanceNorm network?  Candidates, each emulated by rounding exactly the tensors the kernel
would round (products and sums in fp64, so only the operand / storage roundings are visible):

  split16      : bf16x2 today  -- operands carry 16 significant bits, stored tensors 16 bits           (3.0 MFMA / product)
  split22      : f16x2         -- 22 bits                                                              (3.0)
  f16          : single f16    -- 11 bits                                                              (1.0)
  fp8corr      : Wh*xh in f16 + (Wh8*xl8 + Wl8*xh8) with 4-significant-bit (e4m3) operands             (2.0: MX-fp8 at 2x rate)
  wino{1,2,3}  : Winograd F(2,3) along 1 / 2 / 3 axes, transformed operands carry `bits` significant bits
                 (27 -> 18 / 12 / 8 multiplications per output and channel pair: 2.0 / 1.33 / 0.89 MFMA-equivalents with the split)

Run:  python tools/emul_dev_precision.py [size=64] [seed=0]
"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import torch.nn.functional as F

from oracle import unet_ref as R

KW = R.VARIANTS["anatomix-dev"]


def rbits(t, n):
    """Round to n significant bits (round-to-nearest-even on the mantissa), exponent range unlimited."""
    if n is None:
        return t
    m, e = torch.frexp(t)
    return torch.ldexp(torch.round(m * (2.0 ** n)) / (2.0 ** n), e)


def conv_direct(x, w, opbits):
    return R.conv3_reflect(rbits(x, opbits), rbits(w, opbits))


def conv_fp8corr(x, w, _):
    xh, wh = rbits(x, 11), rbits(w, 11)
    xl, wl = x - xh, w - wh
    y = R.conv3_reflect(xh, wh)
    y = y + R.conv3_reflect(rbits(xl, 4), rbits(wh, 4)) + R.conv3_reflect(rbits(xh, 4), rbits(wl, 4))
    return y


def e4m3(t):
    """True OCP e4m3fn rounding with saturation at +-448 (the conversion the kernels use: clamp, then v_cvt_pk_fp8_f32)."""
    return t.clamp(-448.0, 448.0).to(torch.float32).to(torch.float8_e4m3fn).to(t.dtype)


def f16r(t):
    return t.to(torch.float16).to(t.dtype)


def store_f16x2(t):
    h = f16r(t)
    return h + f16r(t - h)


def conv_mx(x, w, _):
    """The arithmetic of precision 'mx': Wh*xh on the f16 MFMA + (Wh8*xl8 + Wl8*xh8) on the block-scaled fp8 MFMA with ONE uniform
    scale: xh8 = e4m3(xh), xl8 = e4m3(2^11 xl), Wh8 = e4m3(2^Sw Wh), Wl8 = e4m3(2^(Sw+11) Wl), Sw from the layer's max |W|."""
    xh, wh = f16r(x), f16r(w)
    xl, wl = x - xh, w - wh
    sw = torch.floor(torch.log2(448.0 / w.abs().max()))
    y = R.conv3_reflect(xh, wh)
    c = R.conv3_reflect(e4m3(xl * 2.0 ** 11), e4m3(wh * 2.0 ** sw)) + R.conv3_reflect(e4m3(xh), e4m3(wl * 2.0 ** (sw + 11)))
    return y + c * 2.0 ** (-(sw + 11))


def conv_mx_part(x, w, _, keep_xl=True, keep_wl=True):
    """conv_mx with one of the two correction products dropped (what a K = 128 MFMA over four taps of ONE kind would compute)."""
    xh, wh = f16r(x), f16r(w)
    xl, wl = x - xh, w - wh
    sw = torch.floor(torch.log2(448.0 / w.abs().max()))
    y = R.conv3_reflect(xh, wh)
    c = 0
    if keep_xl:
        c = c + R.conv3_reflect(e4m3(xl * 2.0 ** 11), e4m3(wh * 2.0 ** sw))
    if keep_wl:
        c = c + R.conv3_reflect(e4m3(xh), e4m3(wl * 2.0 ** (sw + 11)))
    return y + c * 2.0 ** (-(sw + 11))


def e2m3(u):
    """OCP MX fp6 e2m3 grid: steps of 1/8 below 2, 1/4 below 4, 1/2 up to the maximum 7.5 (round to nearest even, saturating)."""
    a = u.abs().clamp(max=7.5)
    step = torch.where(a < 2, 0.125, torch.where(a < 4, 0.25, 0.5)).to(u.dtype)
    return torch.sign(u) * torch.round(a / step) * step


def mx6(t, dim, block=32):
    """Block-scaled e2m3 (MXFP6): blocks of `block` consecutive entries along `dim` share one power-of-two scale 2^(floor(log2 max) - 2)."""
    t = t.movedim(dim, -1)
    shp = t.shape
    b = t.reshape(*shp[:-1], shp[-1] // block, block)
    amax = b.abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(torch.where(amax > 0, amax, torch.ones_like(amax))))
    scale = torch.pow(torch.tensor(2.0, dtype=t.dtype), e - 2)
    q = e2m3(b / scale) * scale
    return q.reshape(shp).movedim(-1, dim)


def conv_mx6(x, w, _, blk=32):
    """Corrections on the MXFP6 (e2m3) MFMA -- twice the fp8 rate on gfx950: Wh*xh on f16 + (Wh6*xl6 + Wl6*xh6), every 32-channel
    block of a voxel / of a (cout, tap) weight row with its own E8M0 scale."""
    xh, wh = f16r(x), f16r(w)
    xl, wl = x - xh, w - wh
    if x.shape[1] % blk:
        return R.conv3_reflect(xh, wh) + R.conv3_reflect(xl, wh) + R.conv3_reflect(xh, wl)      # the stem is not an mx layer
    y = R.conv3_reflect(xh, wh)
    return y + R.conv3_reflect(mx6(xl, 1, blk), mx6(wh, 1, blk)) + R.conv3_reflect(mx6(xh, 1, blk), mx6(wl, 1, blk))


# Winograd F(2,3):  Y = A^T [ (G g G^T) . (B^T d B) ] A
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def _apply(t, mat, dim):
    t = torch.movedim(t, dim, -1)
    t = torch.matmul(t, mat.T.to(t.dtype))
    return torch.movedim(t, -1, dim)


def conv_wino(x, w, bits, axes=3, chunk=8):
    """x [N,C,D,H,W] (fp64), w [O,C,3,3,3].  Winograd along the LAST `axes` spatial axes, direct on the others."""
    N, C, D, H, W = x.shape
    O = w.shape[0]
    xp = F.pad(x, (1, 1, 1, 1, 1, 1), mode="reflect")
    # weight transform (per transformed axis 3 -> 4), rounded to `bits`
    u = w
    for a in range(3 - axes, 3):
        u = _apply(u, G, 2 + a)
    u = rbits(u, bits)
    out = torch.empty(N, O, D, H, W, dtype=x.dtype)
    wino_ax = list(range(3 - axes, 3))
    sizes = [D, H, W]
    # process z in chunks of output planes to bound memory
    zc = chunk if 0 in wino_ax else chunk
    for z0 in range(0, D, zc):
        z1 = min(D, z0 + zc)
        blk = xp[:, :, z0:z1 + 2]                       # padded planes for outputs z0..z1-1
        # unfold the Winograd axes into (tiles, 4) with stride 2; direct axes into (out, 3) with stride 1
        t = blk
        for a in range(3):
            if a in wino_ax:
                t = t.unfold(2 + a, 4, 2)
            else:
                t = t.unfold(2 + a, 3, 1)
        # t: [N, C, nz, ny, nx, kz, ky, kx]
        for a in wino_ax:
            t = _apply(t, BT, 5 + a)
        t = rbits(t.contiguous(), bits)
        m = torch.einsum("ncdhwijk,ocijk->nodhwijk", t, u) if axes == 3 else None
        if axes == 3:
            y = m
            for a in range(3):
                y = _apply(y, AT, 5 + a)
            # y: [N,O,nz,ny,nx,2,2,2] -> interleave
            nz, ny, nx = y.shape[2:5]
            y = y.permute(0, 1, 2, 5, 3, 6, 4, 7).reshape(N, O, nz * 2, ny * 2, nx * 2)
        else:
            # direct axes are summed inside the einsum; Winograd axes keep their transformed index
            letters = "ijk"
            keep = "".join(letters[a] for a in wino_ax)
            y = torch.einsum("ncdhwijk,ocijk->nodhw" + keep, t, u)
            for n_, a in enumerate(wino_ax):
                y = _apply(y, AT, 5 + n_)
            # interleave the transformed axes' (tiles, 2) pairs
            dims = list(y.shape[2:5])
            perm = [0, 1]
            shape = [N, O]
            k = 5
            for a in range(3):
                perm.append(2 + a)
                if a in wino_ax:
                    perm.append(k); k += 1
                    shape.append(dims[a] * 2)
                else:
                    shape.append(dims[a])
            y = y.permute(*perm).reshape(*shape)
        out[:, :, z0:z1] = y
    return out


def forward_emul(x, sd, conv, opbits, storebits):
    kw = dict(ngf=24, norm="batch", final_act="none", activation="relu", pooling="Max", interp="nearest",
              use_skip_connection=True, norm_eps=1e-5, doubleconv=True)
    kw.update(KW)
    p = R.build_plan(**{k: v for k, v in kw.items() if k != "dimension"})
    q = (lambda t: store_f16x2(t)) if storebits == "f16x2" else (lambda t: rbits(t, storebits))
    feat = q(x.double())
    skips = []
    i, n = 0, len(p.kinds)
    while i < n:
        kind = p.kinds[i]
        last = i
        if kind == "conv":
            w = sd[f"model.{i}.weight"].double()
            b = sd[f"model.{i}.bias"].double()
            feat = conv(feat, w, opbits) + b[None, :, None, None, None]
            j = i + 1
            if j < n and p.kinds[j] == "norm":
                feat = F.instance_norm(q(feat), eps=kw["norm_eps"]); j += 1
            if j < n and p.kinds[j] == "act":
                feat = F.relu(feat); j += 1
            if i != max(p.conv_io):
                feat = q(feat)
            last = j - 1
            i = j
        elif kind == "pool":
            feat = q(F.avg_pool3d(feat, 2)); i += 1
        elif kind == "up":
            feat = q(F.interpolate(feat, scale_factor=2, mode="trilinear")); i += 1
        else:
            i += 1
        if last in p.encoder_idx:
            skips.append(feat)
        if last in p.decoder_idx:
            feat = torch.cat((skips.pop(), feat), dim=1)
    return feat


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    which = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    torch.set_num_threads(os.cpu_count())
    sd = R.synthetic_state_dict(KW, seed)
    x = R.synthetic_input(100 + seed, 1, (size,) * 3)
    t0 = time.time()
    ref = forward_emul(x, sd, conv_direct, None, None)
    print("fp64 reference %.1f s" % (time.time() - t0), flush=True)
    ref32 = R.forward(x, sd, KW).double()
    rl2 = lambda a, b: float((a - b).norm() / b.norm())
    mx = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print("fp32 oracle vs fp64 emulation: rel_l2 %.3e" % rl2(ref32, ref), flush=True)
    cases = [
        ("f16 (11/11)", conv_direct, 11, 11),
        ("split16 bf16x2", conv_direct, 16, 16),
        ("split22 f16x2", conv_direct, 22, 22),
        ("fp8corr store15", conv_fp8corr, None, 15),
        ("fp8corr store22", conv_fp8corr, None, 22),
        ("mx e4m3 f16x2store", conv_mx, None, "f16x2"),
        ("mx6 e2m3 block32", conv_mx6, None, "f16x2"),
        ("mx6 e2m3 block16", lambda a, b, c: conv_mx6(a, b, c, 16), None, "f16x2"),
        ("mxpart no Wl", lambda a, b, c: conv_mx_part(a, b, c, True, False), None, "f16x2"),
        ("mxpart no xl", lambda a, b, c: conv_mx_part(a, b, c, False, True), None, "f16x2"),
        ("mxpart neither, f16x2 store", lambda a, b, c: conv_mx_part(a, b, c, False, False), None, "f16x2"),
        ("wino1 16b", lambda a, b, c: conv_wino(a, b, c, 1), 16, 16),
        ("wino2 16b", lambda a, b, c: conv_wino(a, b, c, 2), 16, 16),
        ("wino3 16b", lambda a, b, c: conv_wino(a, b, c, 3), 16, 16),
        ("wino3 22b", lambda a, b, c: conv_wino(a, b, c, 3), 22, 22),
        ("wino3 op22 store16", lambda a, b, c: conv_wino(a, b, c, 3), 22, 16),
        ("wino3 op16 store22", lambda a, b, c: conv_wino(a, b, c, 3), 16, 22),
    ]
    for name, conv, ob, sb in cases:
        if which and not any(k in name for k in which):
            continue
        t0 = time.time()
        y = forward_emul(x, sd, conv, ob, sb)
        print("%-22s rel_l2 %.3e  max_rel %.3e   (%.0f s)" % (name, rl2(y, ref), mx(y, ref), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
