import sys, os, time
sys.path.insert(0, "/root/repo")
import torch, anatomix_amd
from oracle import unet_ref as R
dev = torch.device("cuda:0"); kw = R.VARIANTS["anatomix"]
m = anatomix_amd.Unet(**kw); m.load_state_dict(R.synthetic_state_dict(kw, 0)); m = m.to(dev).eval()
for B in (4, 8):
    x = R.synthetic_input(100, B, (128,)*3).to(dev)
    for cc in (0, 2, 4):
        m.concurrent_chunks = cc
        with torch.no_grad():
            for _ in range(10): y = m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100): y = m(x)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"batch {B} concurrent_chunks {cc}: {B*100/dt:.1f} vol/s", flush=True)
