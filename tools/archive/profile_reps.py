"""Per-rep launch times (hipEvents of the instrumented forward) of the level-0 layers: how stable is roofline.avg_launch_us?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, anatomix_amd
from oracle import unet_ref as R
dev = torch.device("cuda:0"); kw = R.VARIANTS["anatomix"]
m = anatomix_amd.Unet(**kw); m.load_state_dict(R.synthetic_state_dict(kw, 0)); m = m.to(dev).eval()
x = R.synthetic_input(100, 4, (128,) * 3).to(dev)
with torch.no_grad():
    for phase in ("cold", "after 100 forwards", "after 100 forwards + 20 profiled"):
        if phase != "cold":
            for _ in range(100): m(x)
            torch.cuda.synchronize()
        rows = []
        for rep in range(20 if phase.endswith("profiled") else 6):
            _, recs = m.profile_forward(x)
            rows.append([round(r["ms"] * 1e3) for r in recs if r["d"] == 128])
        print(phase)
        for r in rows[-6:]: print("   level-0 launches (us):", r, " sum of all:", None)
