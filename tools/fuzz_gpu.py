"""Randomised parity sweep on the GPU: single convs (amx_conv3d_k3_reflect) and whole networks against the CPU references.
usage: python tools/fuzz_gpu.py [seconds] [seed]"""
import sys, os, time, random, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import anatomix_amd
from _util import run_conv, run_conv_merged, ref_conv, ref_conv_fp64, ref_conv_upcat_merged, rel_l2
from oracle import unet_ref as R
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
nconv = nnet = nbwd = nfail = nreg = nmlp = nvit = 0
import torch.nn.functional as F
from anatomix_amd.model import train_ops as T


def cl(x, dt):
    return x.permute(0, 2, 3, 4, 1).contiguous().to(dt).to(dev)


def ncdhw(x):
    return x.detach().cpu().double().permute(0, 4, 1, 2, 3).contiguous()


def fuzz_regfeat():
    """MIND-SSC / merged pooling / correlation volume at random sizes against the CPU oracle."""
    import numpy as np
    from oracle import registration_ref as RR
    from anatomix_amd.registration import MINDSSC, correlate, smooth_merged_features
    rs = np.random.RandomState(rng.randint(0, 1 << 30))
    shape = tuple(rng.randint(4, m) for m in (20, 24, 40))
    radius, dil = rng.choice([(1, 1), (1, 2), (2, 2), (2, 3), (1, 4)])
    img = rs.rand(*shape).astype(np.float32)
    for _ in range(2):                                               # smooth a little: white noise amplifies fp32 rounding
        img = (img + np.roll(img, 1, 0) + np.roll(img, 1, 1) + np.roll(img, 1, 2)) / 4
    bad = 0
    got = MINDSSC(torch.from_numpy(img).to(dev)[None, None], radius, dil)[0].cpu().numpy()
    e = float(np.abs(got - RR.mindssc(img, radius, dil)).max())
    if not e < 5e-5:
        bad += 1
        print("MIND FAIL", shape, radius, dil, e)
    c, g = rng.randint(1, 20), rng.choice([1, 2, 2, 3, 4])
    if min(shape) >= g:
        feats = rs.randn(c, *shape).astype(np.float32)
        got = smooth_merged_features(torch.from_numpy(RR.mindssc(img, 1, 2)).to(dev)[None], torch.from_numpy(feats).to(dev)[None], g, 0.1)
        want = RR.merged_pooled(RR.mindssc(img, 1, 2), feats, 0.1, g)
        e = float(np.abs(got[0].cpu().numpy() - want).max())
        if not e < 5e-6:
            bad += 1
            print("POOLCAT FAIL", shape, c, g, e)
    hw, ch = rng.choice([1, 1, 2, 3]), rng.randint(1, 30)
    h, w, d = (rng.randint(2, m) for m in (10, 12, 20))
    fix, mov = rs.rand(ch, h, w, d).astype(np.float32), rs.rand(ch, h, w, d).astype(np.float32)
    ssd, amin = correlate(torch.from_numpy(fix).to(dev)[None], torch.from_numpy(mov).to(dev)[None], hw, 1, (h, w, d), ch)
    ref, ref_amin = RR.correlate(fix, mov, hw)
    e = float(np.abs(ssd.cpu().numpy() - ref).max() / np.abs(ref).max())
    if not (e < 1e-5 and (amin.cpu().numpy() == ref_amin).mean() > 0.99):
        bad += 1
        print("CORRELATE FAIL", (ch, h, w, d), hw, e)
    return bad


def fuzz_mlp():
    """Projection head forward + backward at random shapes against float64 modules."""
    import copy
    import torch.nn as nn
    from anatomix_amd.pretraining import mlp_head
    n, cin, width = rng.randint(2, 2048), 4 * rng.randint(1, 64), 8 * rng.randint(1, 40)
    n_mlps, act = rng.choice([2, 3]), rng.choice(["relu", "lrelu"])
    torch.manual_seed(rng.randint(0, 1 << 30))
    A = (lambda: nn.ReLU(inplace=True)) if act == "relu" else (lambda: nn.LeakyReLU(0.3, inplace=True))
    mods = [nn.Linear(cin, width, bias=False), nn.BatchNorm1d(width), A()]
    for _ in range(n_mlps - 2):
        mods += [nn.Linear(width, width, bias=False), nn.BatchNorm1d(width), A()]
    mods += [nn.Linear(width, width, bias=False), nn.BatchNorm1d(width, affine=False)]
    ref = nn.Sequential(*mods).double().train()
    hip = copy.deepcopy(ref).float().to(dev).train()
    x64 = torch.randn(n, cin, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(n, width, dtype=torch.float64)
    y64 = ref(x64)
    (y64 * gy).sum().backward()
    x = x64.detach().float().to(dev).requires_grad_(True)
    y = mlp_head.run_head(hip, x)
    (y * gy.float().to(dev)).sum().backward()
    # L2-relative: a pre-activation within fp32 rounding of zero flips one ReLU mask bit between the fp32 and the fp64 run
    # (~1 run in 15 at these sizes); that changes ONE sample's gradient row by a few per cent -- visible in a max-abs metric,
    # ~1e-3 in L2 -- and is not an error of the kernels
    rel = lambda a, b: ((a.double().cpu() - b).norm() / (b.norm() + 1e-30)).item()
    pairs = [(x.grad, x64.grad)] + [(p.grad, q.grad) for p, q in zip(hip.parameters(), ref.parameters())]
    errs = [rel(y, y64.detach())] + [rel(a, b) for a, b in pairs]
    cond = max(1.0, (64.0 / n) ** 0.5)               # tiny batches make BatchNorm ill-conditioned (rstd ~ 1/sqrt(eps))
    # (a flipped unit also moves its column's BatchNorm sums, i.e. every row a little: ~1/n of the gradient, ~1e-2 in L2 at
    # most; a wrong kernel is an O(1) error) -- forward tight, gradients to 3e-2
    ok = errs[0] < 1e-5 * cond and max(errs) < 3e-2 * cond
    if not ok:
        print("MLP FAIL", dict(n=n, cin=cin, width=width, n_mlps=n_mlps, act=act), max(errs))
        return 1
    return 0

def fuzz_vit():
    """PrimusV2 engine (amx_vit_forward) against the same module composed of torch operators, random constructor arguments inside
    the engine's envelope (token count a multiple of 64, even grid width, head_dim a multiple of 6, embed_dim a multiple of 4)."""
    from anatomix_amd.model.vit3d import PrimusV2
    while True:
        grid = tuple(rng.choice([2, 4, 4, 8]) for _ in range(3))
        if (grid[0] * grid[1] * grid[2]) % 64 == 0:
            break
    heads, hd = rng.choice([(6, 66), (4, 60), (8, 36), (4, 78), (12, 66), (2, 18), (3, 48), (16, 66), (5, 24)])
    kw = dict(input_channels=1, num_classes=rng.choice([4, 16, 32, 36, 64]), embed_dim=heads * hd,
              patch_embed_size=(8, 8, 8), input_shape=tuple(8 * g for g in grid), eva_depth=rng.randint(1, 3), eva_numheads=heads,
              num_register_tokens=rng.choice([0, 1, 3, 8]), init_values=rng.choice([None, 0.1]),
              scale_attn_inner=rng.random() < 0.5, qk_norm=rng.random() < 0.5, out_norm=rng.choice(["none", "demean", "instance"]),
              out_norm_eps=1e-2, in_eps=1e-2)
    torch.manual_seed(rng.randint(0, 1 << 30))
    try:
        m = PrimusV2(**kw).to(dev).eval()
    except (ValueError, NotImplementedError) as ex:                  # a constructor refusal is the documented behaviour
        return 0
    batch = rng.randint(1, 3)
    with torch.no_grad():
        for prm in m.parameters():
            if prm.dim() == 1:
                prm.add_(0.1 * torch.randn_like(prm))
        x = torch.rand(batch, kw["input_channels"], *kw["input_shape"], device=dev)
        y = m(x)
        m.use_engine = False
        y_t = m(x)
    e = rel_l2(y.cpu(), y_t.cpu())
    if not (torch.isfinite(y).all() and e < 1e-3):
        print("VIT FAIL", kw, batch, e)
        return 1
    return 0


devnull = open(os.devnull, "w")
POISON = os.environ.get("AMX_FUZZ_POISON", "0") == "1"   # refill the allocator's free memory with NaNs every few cases
ncase = 0
while time.time() < t_end:
    try:
        ncase += 1
        if POISON and ncase % 5 == 0:
            blocks = [torch.empty(64 << 20, dtype=torch.float32, device=dev).fill_(float("nan")) for _ in range(8)]
            torch.cuda.synchronize()
            del blocks
        pick = rng.random()
        if pick < 0.08:
            nreg += 1
            nfail += fuzz_regfeat()
        elif pick < 0.14:
            nmlp += 1
            nfail += fuzz_mlp()
        elif pick < 0.18:
            nvit += 1
            nfail += fuzz_vit()
        elif pick < 0.3:
            # ---- conv backward: weight gradient + data gradient against torch autograd (double, rounded operands)
            dt = rng.choice([torch.bfloat16, torch.float16])
            up = rng.random() < 0.4
            c0 = rng.choice([16, 32, 64]); c1 = rng.choice([16, 32, 64]) if up else 0
            cin = 1 if (not up and rng.random() < 0.15) else c0 + c1
            cout = rng.choice([16, 32, 64])
            size = tuple(2 * rng.randint(1, m) for m in (5, 7, 30)) if up else tuple(rng.randint(2, m) for m in (9, 13, 70))
            n = rng.randint(1, 2)
            g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
            if cin == 1:
                c0 = 16
            x0 = torch.randn(n, c0 if cin != 1 else 1, *size, generator=g)
            x1 = torch.randn(n, c1, *[s // 2 for s in size], generator=g) if up else None
            w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
            dy = torch.randn(n, cout, *size, generator=g)
            xq = x0.to(dt).double().requires_grad_(True)
            lq = x1.to(dt).double().requires_grad_(True) if up else None
            wq = w.to(dt).double().requires_grad_(True)
            inp = torch.cat((xq, F.interpolate(lq, scale_factor=2, mode="nearest")), 1) if up else xq
            F.conv3d(F.pad(inp, (1,) * 6, mode="reflect"), wq).backward(dy.to(dt).double())
            xd = torch.zeros((n, *size, c0), dtype=dt, device=dev)
            xd[..., : x0.shape[1]] = cl(x0, dt)
            ld = cl(x1, dt) if up else None
            fr = T.new_framed(n, *size, cout, dt, dev)
            T.interior(fr).copy_(cl(dy, dt))
            dw = T.conv_wgrad(fr, xd, ld, cin, cout)
            din = ncdhw(T.conv_dgrad(fr, w.to(dev)))
            ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
            e_w = rel_l2(dw.cpu().double(), wq.grad)
            e_x = rel_l2(din[:, : x0.shape[1]], xq.grad)
            ok = e_w < 2e-5 and e_x < 2 * ulp
            if up:
                lo = [s // 2 for s in size]
                dl = din[:, c0:].reshape(n, c1, lo[0], 2, lo[1], 2, lo[2], 2).sum((3, 5, 7))
                ok = ok and rel_l2(dl, lq.grad) < 2 * ulp
            nbwd += 1
            if not ok:
                nfail += 1
                print("BWD FAIL", dict(dt=str(dt), c0=c0, c1=c1, cin=cin, cout=cout, size=size, n=n), "e_w", e_w, "e_x", e_x)
        elif pick < 0.38:
            # ---- the two-launch merged concat conv (amx_conv3d_upcat_merged), 16-bit and strict precisions, ragged sizes
            prec = rng.choice(["f16", "bf16", "f16x2", "bf16x2"])
            strict = prec.endswith("x2")
            c0 = rng.choice([16, 32, 48, 64] if strict else [32, 48, 64, 128]); c1 = 32 * rng.randint(1, 4)
            size = (2 * rng.randint(2, 6), 2 * rng.randint(2, 9), 2 * rng.randint(16, 36))
            n, act = rng.randint(1, 2), rng.choice([0, 1, 2])
            g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
            x0 = torch.randn(n, c0, *size, generator=g)
            x1 = torch.randn(n, c1, *[s // 2 for s in size], generator=g)
            w = torch.randn(c0, c0 + c1, 3, 3, 3, generator=g) / (27 * (c0 + c1)) ** 0.5
            scale = (0.5 + torch.rand(c0, generator=g)) if rng.random() < 0.5 else None
            shift = torch.randn(c0, generator=g) * 0.1 if rng.random() < 0.7 else None
            y = run_conv_merged(dev, x0, x1, w, scale, shift, act, prec)
            if strict:
                r, tol = ref_conv_fp64(x0, x1, w, scale, shift, act), (4e-5 if prec == "bf16x2" else 6e-6)
            else:
                r, tol = ref_conv_upcat_merged(x0, x1, w, scale, shift, act, prec, round_partial=True), (4e-4 if prec == "f16" else 3.2e-3)
            e = rel_l2(y, r)
            nconv += 1
            if not (e < tol) or torch.isnan(y).any():
                nfail += 1
                print("MERGED CONV FAIL", dict(prec=prec, c0=c0, c1=c1, size=size, n=n, act=act), "rel_l2", e)
        elif pick < 0.7:
            prec = rng.choice(["f16", "bf16"])
            up = rng.random() < 0.35
            c0 = rng.choice([16, 32, 48, 64, 128]); c1 = rng.choice([16, 32, 64, 128]) if up else 0
            cout = rng.choice([16, 32, 48, 64, 96, 128])
            if up:
                size = tuple(2 * rng.randint(1, m) for m in (6, 8, 20))
            else:
                size = tuple(rng.randint(2, m) for m in (12, 16, 40))
            n = rng.randint(1, 3)
            act = rng.choice([0, 1, 2])
            g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
            x0 = torch.randn(n, c0, *size, generator=g)
            x1 = torch.randn(n, c1, *[s // 2 for s in size], generator=g) if up else None
            w = torch.randn(cout, c0 + c1, 3, 3, 3, generator=g) / (27 * (c0 + c1)) ** 0.5
            scale = (0.5 + torch.rand(cout, generator=g)) if rng.random() < 0.5 else None
            shift = torch.randn(cout, generator=g) * 0.1 if rng.random() < 0.7 else None
            planar = (cout <= 32 and size[2] >= 32 and rng.random() < 0.3)
            y = run_conv(dev, x0, x1, w, scale, shift, act, prec, planar=planar)
            merged = up and c0 == 16 and c1 == 32 and cout == 16 and size[2] >= 32 and size[1] >= 8 and size[0] >= 4
            r = (ref_conv_upcat_merged if merged else ref_conv)(x0, x1, w, scale, shift, act, prec)
            tol = (2.0 ** -11 if prec == "f16" else 2.0 ** -8) * (0.02 if planar else 1.0) + 2e-6
            e = rel_l2(y, r)
            nconv += 1
            if not (e < tol) or torch.isnan(y).any():
                nfail += 1
                print("CONV FAIL", dict(prec=prec, c0=c0, c1=c1, cout=cout, size=size, n=n, act=act, planar=planar, scale=scale is not None), "rel_l2", e)
        else:
            nd = rng.randint(1, 3)
            kw = dict(dimension=3, input_nc=rng.choice([1, 1, 1, 2, 3]), output_nc=rng.choice([16, 32, 16, 32, 5, 33]), num_downs=nd, ngf=rng.choice([8, 16, 24, 32]),
                      norm=rng.choice(["batch", "batch", "instance", "instance_affine", "none"]), activation=rng.choice(["relu", "lrelu"]),
                      pooling=rng.choice(["Max", "Avg"]), interp=rng.choice(["nearest", "trilinear"]),
                      doubleconv=rng.random() < 0.8, use_skip_connection=rng.random() < 0.85)
            if kw["norm"] != "batch":
                kw["norm_eps"] = 1e-2
            mul = 1 << nd
            size = tuple(mul * rng.randint(2 if a < 2 else max(2, 32 // mul), 6 if a < 2 else max(3, 64 // mul)) for a in range(3))
            so, sys.stdout = sys.stdout, devnull
            m = anatomix_amd.Unet(**kw)
            sys.stdout = so
            sd = R.synthetic_state_dict(kw, rng.randint(0, 99))
            m.load_state_dict(sd, strict=True)
            m = m.to(dev).eval()
            nb = rng.randint(1, 2)
            x = torch.cat([R.synthetic_input(rng.randint(0, 999), nb, size) for _ in range(kw["input_nc"])], dim=1)
            nmod = len(m.model)
            layers = sorted(rng.sample(range(nmod), rng.randint(0, 4)))
            with torch.no_grad():
                if layers:
                    y, feats = m(x.to(dev), layers)
                    ry, rfeats = R.forward(x, sd, kw, layers=layers)
                else:
                    y, feats, rfeats = m(x.to(dev)), [], []
                    ry = R.forward(x, sd, kw)
                rl = R.forward_lowp(x, sd, kw, torch.float16)
            e_ref, e_emu = rel_l2(y.cpu(), ry), rel_l2(rl, ry)
            nnet += 1
            ok = e_ref < max(2.5 * e_emu, 2e-3) and all(a.shape == b.shape for a, b in zip(feats, rfeats))
            for l, a, b in zip(layers, feats, rfeats):
                ok = ok and rel_l2(a.cpu(), b) < max(4 * e_emu, 4e-3)
            if not ok:
                nfail += 1
                print("NET FAIL", kw, size, layers, "e_ref", e_ref, "e_emul", e_emu, [rel_l2(a.cpu(), b) for a, b in zip(feats, rfeats)])
    except Exception as ex:
        nfail += 1
        print("EXCEPTION", type(ex).__name__, str(ex)[:300])
        traceback.print_exc(limit=2)
print(f"fuzz: {nconv} convs, {nbwd} conv backwards, {nnet} networks, {nreg} registration-feature cases, {nmlp} projection heads, {nvit} ViT configurations, "
      f"{nfail} failures in {budget:.0f} s")
