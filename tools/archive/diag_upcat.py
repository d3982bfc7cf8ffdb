import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from _util import run_conv, ref_conv, ref_conv_upcat_merged, rel_l2, max_rel
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
d, h, w = 16, 16, 32
for mode in ["both", "skip_only", "up_only"]:
    x0 = torch.from_numpy(rs.randn(1, 16, d, h, w).astype(np.float32))
    x1 = torch.from_numpy(rs.randn(1, 32, d // 2, h // 2, w // 2).astype(np.float32))
    wt = torch.from_numpy((rs.randn(16, 48, 3, 3, 3) / 36).astype(np.float32))
    if mode == "skip_only": wt[:, 16:] = 0
    if mode == "up_only": wt[:, :16] = 0
    got = run_conv(dev, x0, x1, wt, None, None, 0, "f16", planar=True)
    ref = ref_conv_upcat_merged(x0, x1, wt, None, None, 0, "f16")
    err = (got - ref).abs()
    print(mode, "rel_l2 %.3e max_rel %.3e" % (rel_l2(got, ref), max_rel(got, ref)))
    if max_rel(got, ref) > 1e-3:
        e = err[0].amax(0)   # [d,h,w]
        for pz in (0, 1):
            for py in (0, 1):
                for px in (0, 1):
                    print("  parity", pz, py, px, "max err %.3e" % e[pz::2, py::2, px::2].max().item())
        print("  err by z:", [round(v, 3) for v in e.amax((1, 2)).tolist()])
        print("  err by y:", [round(v, 3) for v in e.amax((0, 2)).tolist()])
        print("  err by x:", [round(v, 3) for v in e.amax((0, 1)).tolist()])
        print("  err by channel:", [round(v, 3) for v in err[0].amax((1, 2, 3)).tolist()])
