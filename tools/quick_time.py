"""Scratch timing of the forward (not the bench contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anatomix_amd
from oracle import unet_ref as R
dev = torch.device("cuda:0")
kw = R.VARIANTS["anatomix"]
m = anatomix_amd.Unet(**kw); m.load_state_dict(R.synthetic_state_dict(kw, 0)); m = m.to(dev).eval()
for n in (1, 2):
    x = R.synthetic_input(100, n, (128, 128, 128)).to(dev)
    with torch.no_grad():
        for _ in range(3): y = m(x)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        K = 20
        for _ in range(K): y = m(x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
    print(f"N={n}: {ms:.3f} ms/forward -> {n/ms*1e3:.1f} vol/s, {346.99*n/ms:.1f} TFLOP/s")
