"""Host-side enqueue time of the training forward / backward (GPU idle at start, no sync inside)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import anatomix_amd
from oracle import unet_ref as R, pretrain_inputs as PI
dev = torch.device("cuda:0")
kw = R.VARIANTS["anatomix"]
so = sys.stdout; sys.stdout = open(os.devnull, "w")
netG = anatomix_amd.Unet(**kw); netG.load_state_dict(R.synthetic_state_dict(kw, 3, gain=2 ** 0.5)); netG.precision = "bf16"
netG = netG.to(dev).train()
sys.stdout = so
A, B, seg = [t.to(dev) for t in PI.step_inputs(128)]
x = torch.cat((A, B))
for _ in range(3):
    out, feats = netG(x, PI.NCE_LAYERS); (out.mean() + sum(f.mean() for f in feats)).backward()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out, feats = netG(x, PI.NCE_LAYERS)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    loss = out.mean() + sum(f.mean() for f in feats)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter(); torch.cuda.synchronize(); t5 = time.perf_counter()
    print(f"forward: host {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms | backward: host {1e3*(t4-t3):.2f} ms, total {1e3*(t5-t3):.2f} ms")
