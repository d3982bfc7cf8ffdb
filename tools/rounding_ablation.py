"""Which layers' f16 rounding makes the max-norm outliers of the f16 headline forward?  
CPU only: oracle.unet_ref.forward_lowp with the rounding points of a chosen set of conv layers switched off (their folded weights and
their stored output stay fp32 -- what keeping that layer in the f16x2 pair format would give, to first order).  For each weight seed:
the all-f16 distances from the fp32 oracle, the gain of exempting each single layer, and a greedy set until max-rel <= 1e-3.
usage: python tools/rounding_ablation.py [size=64] [seeds=0,1,2,3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import unet_ref as R

KW = R.VARIANTS["anatomix"]


def forward_mixed(x, sd, kwargs, exact=frozenset(), lowp=torch.float16, exact_input=False):
    kw = dict(ngf=24, norm="batch", final_act="none", activation="relu", pooling="Max", interp="nearest",
              use_skip_connection=True, norm_eps=1e-5, doubleconv=True)
    kw.update(kwargs)
    p = R.build_plan(**{k: v for k, v in kw.items() if k != "dimension"})
    q = lambda t: t.to(lowp).to(torch.float32)
    feat = x.float() if exact_input else q(x.float())
    skips = []
    i, n = 0, len(p.kinds)
    while i < n:
        kind = p.kinds[i]
        last = i
        if kind == "conv":
            w, t = R.fold_conv_params(sd, kw, i, p)
            ex = i in exact
            feat = R.conv3_reflect(feat, w.float() if ex else q(w.float()), t.float())
            j = i + 1
            if j < n and p.kinds[j] == "norm":
                j += 1
            if j < n and p.kinds[j] == "act":
                feat = R.act_apply(feat, kw["activation"]); j += 1
            if i != max(p.conv_io) and not ex:
                feat = q(feat)
            last = j - 1
            i = j
        elif kind == "pool":
            feat = F.max_pool3d(feat, 2); i += 1
        elif kind == "up":
            feat = F.interpolate(feat, scale_factor=2, mode="nearest"); i += 1
        else:
            i += 1
        if last in p.encoder_idx:
            skips.append(feat)
        if last in p.decoder_idx:
            feat = torch.cat((skips.pop(), feat), dim=1)
    return feat, [k for k in range(n) if p.kinds[k] == "conv"]


def dist(y, ref):
    d = (y - ref).double()
    return float(d.norm() / ref.double().norm()), float(d.abs().max() / ref.abs().max())


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    seeds = [int(s) for s in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2,3".split(","))]
    torch.set_num_threads(8)
    for seed in seeds:
        sd = R.synthetic_state_dict(KW, seed)
        x = R.synthetic_input(100 + seed, 1, (size,) * 3)
        with torch.no_grad():
            ref = R.forward(x, sd, KW)
            y0, convs = forward_mixed(x, sd, KW)
            l2, mx = dist(y0, ref)
            print(f"seed {seed} size {size}: all f16  rel-L2 {l2:.3e}  max-rel {mx:.3e}", flush=True)
            gains = []
            for c in convs:
                y, _ = forward_mixed(x, sd, KW, exact=frozenset([c]))
                a, b = dist(y, ref)
                gains.append((b, a, c))
                print(f"   layer m{c:2d} exempt: rel-L2 {a:.3e}  max-rel {b:.3e}", flush=True)
            yi, _ = forward_mixed(x, sd, KW, exact_input=True)
            print(f"   input exempt:    rel-L2 %.3e  max-rel %.3e" % dist(yi, ref), flush=True)
            chosen = []
            cur = mx
            while cur > 1e-3 and len(chosen) < 6:
                best = None
                for c in convs:
                    if c in chosen:
                        continue
                    y, _ = forward_mixed(x, sd, KW, exact=frozenset(chosen + [c]))
                    a, b = dist(y, ref)
                    if best is None or b < best[0]:
                        best = (b, a, c)
                chosen.append(best[2]); cur = best[0]
                print(f"   greedy + m{best[2]}: set {chosen}  rel-L2 {best[1]:.3e}  max-rel {best[0]:.3e}", flush=True)


if __name__ == "__main__":
    main()
