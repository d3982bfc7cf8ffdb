"""Randomised check of the weight-gradient kernel (amx_conv3d_wgrad) against torch autograd in double on the rounded operands:
ragged sizes over every geometry class (W <= 8 / 16 / 32 / 128), both storage types, plain and upsample-concat inputs, 1-channel stem.
usage: python tools/wgrad_fuzz.py [seconds] [seed]"""
import sys, os, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from anatomix_amd.model import train_ops as T
from _util import rel_l2
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_case = n_fail = 0
worst = 0.0


def cl(x, dt):
    return x.permute(0, 2, 3, 4, 1).contiguous().to(dt).to(dev)


while time.time() < t_end:
    dt = rng.choice([torch.bfloat16, torch.float16])
    up = rng.random() < 0.4
    c0 = rng.choice([16, 32, 64]); c1 = rng.choice([16, 32, 48]) if up else 0
    cin = 1 if (not up and rng.random() < 0.15) else c0 + c1
    cout = rng.choice([16, 32, 48])
    wmax = rng.choice([8, 16, 32, 64, 128])
    if up:
        size = (2 * rng.randint(1, 6), 2 * rng.randint(1, 12), 2 * rng.randint(max(1, wmax // 4), wmax // 2))
    else:
        size = (rng.randint(2, 12), rng.randint(2, 24), rng.randint(max(2, wmax // 2), wmax))
    n = rng.randint(1, 3)
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    if cin == 1:
        c0 = 16
    x0 = torch.randn(n, c0 if cin != 1 else 1, *size, generator=g)
    x1 = torch.randn(n, c1, *[s // 2 for s in size], generator=g) if up else None
    dy = torch.randn(n, cout, *size, generator=g)
    xq = x0.to(dt).double().to(dev)
    inp = torch.cat((xq, F.interpolate(x1.to(dt).double().to(dev), scale_factor=2, mode="nearest")), 1) if up else xq
    wq = torch.zeros(cout, inp.shape[1], 3, 3, 3, dtype=torch.float64, device=dev, requires_grad=True)
    F.conv3d(F.pad(inp, (1,) * 6, mode="reflect"), wq).backward(dy.to(dt).double().to(dev))
    xd = torch.zeros((n, *size, c0), dtype=dt, device=dev)
    xd[..., : x0.shape[1]] = cl(x0, dt)
    ld = cl(x1, dt) if up else None
    fr = T.new_framed(n, *size, cout, dt, dev)
    T.interior(fr).copy_(cl(dy, dt))
    dw = T.conv_wgrad(fr, xd, ld, cin, cout)
    dw2 = T.conv_wgrad(fr, xd, ld, cin, cout)
    base = torch.randn_like(dw)
    dw3 = T.conv_wgrad(fr, xd, ld, cin, cout, out=base.clone(), accumulate=True)
    e = max(rel_l2(dw.double(), wq.grad), rel_l2(dw3.double(), base.double() + wq.grad))
    n_case += 1
    worst = max(worst, e)
    if not (e < 2e-5) or not torch.equal(dw, dw2):
        n_fail += 1
        print("WGRAD FAIL", dict(dt=str(dt), c0=c0, c1=c1, cin=cin, cout=cout, size=size, n=n), "err", e, "deterministic", torch.equal(dw, dw2), flush=True)
print(f"wgrad fuzz: {n_case} cases, {n_fail} failures, worst rel-L2 {worst:.2e}")
sys.exit(1 if n_fail else 0)
