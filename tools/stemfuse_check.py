"""Stem-fed z-march (modules 0..5 as one launch) against the unfused route (a tap request disables the fusion): must be bit-identical."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anatomix_amd
from oracle import unet_ref as R
dev = torch.device("cuda:0")
kw = R.VARIANTS["anatomix"]
ok = True
for prec in ("f16", "bf16"):
    m = anatomix_amd.Unet(**kw); m.load_state_dict(R.synthetic_state_dict(kw, 0)); m.precision = prec; m = m.to(dev).eval()
    for (n, shp) in [(1, (32, 32, 32)), (2, (64, 64, 64)), (1, (48, 64, 96)), (4, (128, 128, 128)), (1, (16, 32, 64)), (3, (80, 40, 32))]:
        x = R.synthetic_input(100 + n, n, shp).to(dev)
        with torch.no_grad():
            y = m(x)
            y2, feats = m.forward_hip_taps(x, [2])
        torch.cuda.synchronize()
        same = torch.equal(y, y2)
        d = (y - y2).abs().max().item()
        print(prec, n, shp, "bit-identical" if same else f"DIFF max {d:.3e}", "finite" if torch.isfinite(y).all() else "NONFINITE", flush=True)
        ok &= same
print("ALL OK" if ok else "FAILED")
