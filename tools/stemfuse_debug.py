"""Where does the stem-fed z-march differ from stem + conv?  Compares the tap at module 5 (fused) with the same tap of the unfused route."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anatomix_amd
from oracle import unet_ref as R
dev = torch.device("cuda:0")
kw = R.VARIANTS["anatomix"]
m = anatomix_amd.Unet(**kw); m.load_state_dict(R.synthetic_state_dict(kw, 0)); m.precision = "f16"; m = m.to(dev).eval()
n, shp = 1, tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 32, 64)
x = R.synthetic_input(101, n, shp).to(dev)
with torch.no_grad():
    _, fa = m.forward_hip_taps(x, [5])
    _, fb = m.forward_hip_taps(x, [2, 5])
a, b = fa[0], fb[1]
d = (a - b).abs()
print("max diff", d.max().item(), "ref max", b.abs().max().item())
bad = d > 1e-6
print("bad voxels", bad.any(dim=1).sum().item(), "of", bad[:, 0].numel())
bz = bad.any(dim=1)[0]
print("bad per z:", bz.sum(dim=(1, 2)).tolist())
print("bad per y:", bz.sum(dim=(0, 2)).tolist())
print("bad per x:", bz.sum(dim=(0, 1)).tolist())
