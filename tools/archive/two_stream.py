"""Experiment: two model replicas on two HIP streams (batch split) vs one stream -- do the underfilled deep-level
kernels of one forward overlap with the HBM-bound level-0 kernels of the other?"""
import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anatomix_amd
from oracle import unet_ref as R
dev = torch.device("cuda:0")
kw = R.VARIANTS["anatomix"]
def mk():
    m = anatomix_amd.Unet(**kw); m.load_state_dict(R.synthetic_state_dict(kw, 0)); return m.to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
models = [mk() for _ in range(NS)]
streams = [torch.cuda.Stream(dev) for _ in range(NS)]
xs = [R.synthetic_input(100 + i, B // NS, (128,) * 3).to(dev) for i in range(NS)]
xall = R.synthetic_input(100, B, (128,) * 3).to(dev)
def run_split(steps):
    for _ in range(steps):
        for m, s, x in zip(models, streams, xs):
            with torch.cuda.stream(s):
                m(x)
def run_one(steps):
    for _ in range(steps):
        models[0](xall)
with torch.no_grad():
    for fn, name in ((run_one, "1 stream  B=%d" % B), (run_split, "%d streams B=%d each" % (NS, B // NS))):
        fn(5); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(50); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name}: {dt/50*1e3:.3f} ms/step  {B*50/dt:.1f} vol/s")
