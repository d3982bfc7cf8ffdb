import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from anatomix_amd.model.load_from_hf import build_variant
from oracle import unet_ref as R
dev = torch.device("cuda:0")
kw = R.VARIANTS["anatomix-dev"]
m = build_variant("anatomix-dev"); m.load_state_dict(R.synthetic_state_dict(kw, 0)); m = m.to(dev).eval()
for prec in ("f16", "strict"):
    m.precision = prec
    for B, c in ((2, 0), (4, 0), (4, 2), (8, 0), (8, 4), (8, 2)):
        x = R.synthetic_input(100, B, (128,) * 3).to(dev)
        m.concurrent_chunks = c
        with torch.no_grad():
            for _ in range(3): m(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            K = 10
            for _ in range(K): m(x)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"dev {prec} batch {B} chunks {c}: {B*K/dt:.1f} vol/s")
