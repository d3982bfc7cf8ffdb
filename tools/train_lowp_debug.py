"""Where does the HIP training forward leave the rounding-point emulation?  Pre-norm taps at every conv id, f16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import anatomix_amd
from oracle import unet_ref as R, train_lowp as TL
KW = R.VARIANTS["anatomix"]
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
gain = float(sys.argv[2]) if len(sys.argv) > 2 else 2 ** 0.5
dev = torch.device("cuda:0")
hip = anatomix_amd.Unet(**KW); hip.load_state_dict(R.synthetic_state_dict(KW, 3, gain=gain)); hip.precision = prec; hip = hip.to(dev).train()
sd = {k: v.detach().cpu().clone() for k, v in hip.state_dict().items()}
layers = [0, 3, 6, 10, 13, 17, 20, 24, 27, 31, 34, 38, 41, 45, 48, 52, 55, 59, 62]
x = torch.from_numpy(np.random.RandomState(3).rand(2, 1, 64, 64, 64).astype(np.float32))
with torch.no_grad():
    out, feats = hip(x.to(dev), layers)
dt = torch.float16 if prec == "f16" else torch.bfloat16
params = {k: v.float() for k, v in sd.items() if v.dtype.is_floating_point}
with torch.no_grad():
    out_r, taps_r = TL.forward_train_lowp(x, params, KW, layers, dt)
for l, f, t in zip(layers, feats, taps_r):
    d = (f.cpu().double() - t.double())
    print(f"m{l:2d} rel-L2 {float(d.norm() / t.double().norm()):.2e}  max {float(d.abs().max() / t.abs().max()):.2e}  frac differing {float((d != 0).float().mean()):.3f}")
print("out", float((out.cpu().double() - out_r.double()).norm() / out_r.double().norm()))
