"""Scratch: per-layer accuracy of the MFMA conv vs an fp64 reference on identical rounded operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from _util import run_conv, ref_conv, rel_l2, max_rel
import anatomix_amd
from oracle import unet_ref as R
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
for (c0, cout, S) in [(16, 16, 32), (64, 16, 32), (256, 16, 32)]:
    x0 = torch.from_numpy(rs.randn(1, c0, 8, 8, S).astype(np.float32))
    w = torch.from_numpy((rs.randn(cout, c0, 3, 3, 3) / np.sqrt(27.0 * c0)).astype(np.float32))
    got = run_conv(dev, x0, None, w, None, None, 0, "f16", planar=True)
    ref = ref_conv(x0, None, w, None, None, 0, "f16")          # fp64 accumulate
    x32 = x0.half().float(); w32 = w.half().float()
    cpu32 = torch.nn.functional.conv3d(torch.nn.functional.pad(x32, (1,)*6, mode="reflect"), w32)
    print(f"cin={c0}: hip-vs-fp64 rel_l2 {rel_l2(got, ref):.2e} max {max_rel(got, ref):.2e} | cpu fp32-vs-fp64 {rel_l2(cpu32, ref):.2e}")
kw = R.VARIANTS["anatomix"]
for gain in (1.0, 2 ** 0.5):
    sd = R.synthetic_state_dict(kw, 0, gain=gain)
    m = anatomix_amd.Unet(**kw); m.load_state_dict(sd); m = m.to(dev).eval()
    x = R.synthetic_input(100, 1, (64, 64, 64))
    with torch.no_grad():
        y = m(x.to(dev)).cpu(); ref = R.forward(x, sd, kw, dtype=torch.float64); em = R.forward_lowp(x, sd, kw)
    print(f"gain {gain:.3f}: hip-vs-fp64 {rel_l2(y, ref):.2e} (max {max_rel(y, ref):.2e}) | emul-vs-fp64 {rel_l2(em, ref):.2e} | hip-vs-emul {rel_l2(y, em):.2e}")
