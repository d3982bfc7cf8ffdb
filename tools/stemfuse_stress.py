"""Race hunt: the same forward many times, every output compared bit for bit with the first (and with the two-launch route)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anatomix_amd
from oracle import unet_ref as R
dev = torch.device("cuda:0")
kw = R.VARIANTS["anatomix"]
m = anatomix_amd.Unet(**kw); m.load_state_dict(R.synthetic_state_dict(kw, 0)); m.precision = "f16"; m = m.to(dev).eval()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for (n, shp) in [(1, (128, 128, 128)), (2, (128, 128, 128)), (4, (128, 128, 128)), (1, (64, 64, 64)), (3, (32, 64, 96))]:
    x = R.synthetic_input(100 + n, n, shp).to(dev)
    with torch.no_grad():
        ref, _ = m.forward_hip_taps(x, [2])
        bad = 0
        for i in range(reps):
            y = m(x)
            if not torch.equal(y, ref):
                bad += 1
                d = (y - ref).abs()
                if bad <= 2:
                    nz = (d > 0).any(dim=1)
                    print("   mismatch: max", d.max().item(), "voxels", nz.sum().item(), "z", sorted(set(nz.nonzero()[:, 1].tolist()))[:12],
                          "y", sorted(set(nz.nonzero()[:, 2].tolist()))[:12], "x", sorted(set(nz.nonzero()[:, 3].tolist()))[:12])
    print(n, shp, "mismatching runs:", bad, "of", reps, flush=True)
