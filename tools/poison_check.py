"""Runs small / unusual inference configurations after poisoning the caching allocator's free memory with NaNs or large random
values: a kernel that reads memory nobody wrote shows up as a wrong or non-finite result (fresh pages are zero and hide it)."""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import anatomix_amd
from _util import rel_l2
from oracle import unet_ref as R
dev = torch.device("cuda:0")


def poison(kind):
    blocks = [torch.empty(64 << 20, dtype=torch.float32, device=dev) for _ in range(12)]      # 3 GB
    for b in blocks:
        if kind == "nan":
            b.fill_(float("nan"))
        else:
            b.uniform_(-1e4, 1e4)
    torch.cuda.synchronize()
    del blocks


CASES = [
    (dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=16), (8, 12, 16)),
    (dict(dimension=3, input_nc=1, output_nc=64, num_downs=1, ngf=16), (32, 32, 32)),
    (dict(dimension=3, input_nc=1, output_nc=16, num_downs=3, ngf=16, doubleconv=False), (16, 16, 24)),
    (dict(dimension=3, input_nc=1, output_nc=16, num_downs=4, ngf=16), (32, 32, 32)),
    (dict(dimension=3, input_nc=1, output_nc=32, num_downs=2, ngf=16), (16, 16, 16)),       # export path, output_nc > ngf, W < 32
    (dict(dimension=3, input_nc=1, output_nc=128, num_downs=1, ngf=16, use_skip_connection=False), (32, 32, 64)),
    (dict(dimension=3, input_nc=1, output_nc=32, num_downs=5, ngf=32, norm="instance", pooling="Avg", interp="trilinear", norm_eps=1e-2), (64, 64, 64)),
]
bad = 0
for kind in ("nan", "rand", "nan"):
    for kw, size in CASES:
        with contextlib.redirect_stdout(io.StringIO()):
            m = anatomix_amd.Unet(**kw)
        sd = R.synthetic_state_dict(kw, 4)
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).eval()
        x = R.synthetic_input(21, 2, size)
        ref = R.forward(x, sd, kw)
        for rep in range(3):
            poison(kind)
            with torch.no_grad():
                y = m(x.to(dev)).cpu()
            e = rel_l2(y, ref)
            ok = bool(torch.isfinite(y).all()) and e < 2e-2
            if rep == 0 and kw.get("norm", "batch") == "batch":            # the feature-tap forward shares the arena
                nmod = len(m.model)
                layers = [0, nmod // 3, nmod // 2, nmod - 1]
                poison(kind)
                with torch.no_grad():
                    yt, feats = m(x.to(dev), layers)
                    _, rfeats = R.forward(x, sd, kw, layers=layers)
                et = max([rel_l2(yt.cpu(), ref)] + [rel_l2(a.cpu(), b) for a, b in zip(feats, rfeats)])
                ok = ok and et < 2e-2
                e = max(e, et)
            if not ok:
                bad += 1
                print("BAD", kind, kw.get("output_nc"), kw.get("num_downs"), size, "rep", rep, "rel_l2", e, "finite", bool(torch.isfinite(y).all()), flush=True)
print("poison check:", bad, "bad results")
