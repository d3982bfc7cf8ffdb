"""Per-launch timing table of one forward (hipEvents via amx_unet_forward_profiled)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anatomix_amd
from oracle import unet_ref as R
dev = torch.device("cuda:0")
variant = sys.argv[1] if len(sys.argv) > 1 else "anatomix"
batches = [int(a) for a in sys.argv[2:]] or [1, 2]
kw = R.VARIANTS[variant]
m = anatomix_amd.Unet(**kw); m.load_state_dict(R.synthetic_state_dict(kw, 0)); m = m.to(dev).eval()
def call(f):
    """Timing ablations (AMX_ZX_DBG ...) compute garbage: swallow the overflow guard's report of the previous forward and go on."""
    for _ in range(4):
        try:
            return f()
        except Exception as e:
            if "outside the f16 range" not in str(e) or not os.environ.get("LP_IGNORE_OVERFLOW"):
                raise
            torch.cuda.synchronize()
    raise RuntimeError("overflow report repeated")


for n in batches:
    x = R.synthetic_input(100, n, (128, 128, 128)).to(dev)
    with torch.no_grad():
        for _ in range(3): call(lambda: m(x))
        acc = None
        reps = 5
        for _ in range(reps):
            _, recs = call(lambda: m.profile_forward(x))
            if acc is None: acc = [dict(r) for r in recs]
            else:
                for a, r in zip(acc, recs): a["ms"] += r["ms"]
    tot = sum(a["ms"] for a in acc) / reps
    print(f"--- N={n}: sum of launches {tot*1e3:.1f} us")
    for a in acc:
        us = a["ms"] / reps * 1e3
        tf = a["flops"] / (us * 1e-6) / 1e12 if a["flops"] else 0
        gbs = a["bytes"] / (us * 1e-6) / 1e9
        print(f"  m{a['module_idx']:2d} {a['kernel']:42s} {a['cin']:4d}->{a['cout']:3d} @{a['w']:3d} {us:8.1f} us {tf:7.1f} TF {gbs:7.0f} GB/s")
