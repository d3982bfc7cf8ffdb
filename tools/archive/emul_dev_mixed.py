"""CPU emulation (round 5): anatomix-dev with the arithmetic chosen PER RESOLUTION LEVEL -- plain f16 operands / f16 storage at the
shallow levels (where InstanceNorm averages over 10^5..10^6 voxels and the 6 M network shows f16 is enough), the compliant
f16x2mx arithmetic only where the norm planes are small.  Question: does the network still hold 1e-3 against fp32, and from which level on
is the split needed?   python tools/emul_dev_mixed.py [size=64] [seed=0]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import torch.nn.functional as F
from oracle import unet_ref as R
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import emul_dev_precision as E

KW = E.KW


def forward_mixed(x, sd, first_split_level, size, split_conv=E.conv_mx, lowp_store_bits=11, stem_split=True):
    kw = dict(ngf=24, norm="batch", final_act="none", activation="relu", pooling="Max", interp="nearest",
              use_skip_connection=True, norm_eps=1e-5, doubleconv=True)
    kw.update(KW)
    p = R.build_plan(**{k: v for k, v in kw.items() if k != "dimension"})
    qs = E.store_f16x2
    ql = lambda t: E.rbits(t, lowp_store_bits)
    level = lambda t: int(round(torch.log2(torch.tensor(size / t.shape[-1])).item()))
    def q(t, lv=None):
        lv = level(t) if lv is None else lv
        return qs(t) if lv >= first_split_level else ql(t)
    feat = q(x.double())
    skips = []
    i, n = 0, len(p.kinds)
    first = True
    while i < n:
        kind = p.kinds[i]
        last = i
        if kind == "conv":
            w = sd[f"model.{i}.weight"].double()
            b = sd[f"model.{i}.bias"].double()
            lv = level(feat)
            if lv >= first_split_level or (first and stem_split):
                y = split_conv(feat, w, None)
            else:
                y = E.conv_direct(feat, w, 11)
            first = False
            feat = y + b[None, :, None, None, None]
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
            feat = F.avg_pool3d(feat, 2); feat = q(feat); i += 1
        elif kind == "up":
            feat = F.interpolate(feat, scale_factor=2, mode="trilinear"); feat = q(feat); i += 1
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
    torch.set_num_threads(os.cpu_count())
    sd = R.synthetic_state_dict(KW, seed)
    x = R.synthetic_input(100 + seed, 1, (size,) * 3)
    ref = E.forward_emul(x, sd, E.conv_direct, None, None)
    rl2 = lambda a, b: float((a - b).norm() / b.norm())
    mx = lambda a, b: float((a - b).abs().max() / b.abs().max())
    for fsl in (0, 1, 2, 3, 6):
        for stem_split in (True, False):
            if fsl == 0 and not stem_split:
                continue
            t0 = time.time()
            y = forward_mixed(x, sd, fsl, size, stem_split=stem_split)
            print(f"split from level {fsl} (plain f16 above it), stem {'split' if stem_split else 'f16'}: rel_l2 {rl2(y, ref):.3e}  max_rel {mx(y, ref):.3e}  ({time.time()-t0:.0f} s)",
                  flush=True)


if __name__ == "__main__":
    main()
