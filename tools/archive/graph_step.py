"""Experiment: capture the contrastive step (forward + backward [+ AdamW]) in a HIP graph and replay it."""
import contextlib, io, os, sys, time
from argparse import Namespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anatomix_amd
from anatomix_amd.pretraining import PatchSampleF, SupPatchNCELoss
from oracle import unet_ref as R, pretrain_inputs as PI

stage = sys.argv[1] if len(sys.argv) > 1 else "unet"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
kw = R.VARIANTS["anatomix"]
with contextlib.redirect_stdout(io.StringIO()):
    netG = anatomix_amd.Unet(**kw)
    netG.load_state_dict(R.synthetic_state_dict(kw, 3, gain=2 ** 0.5))
    netF = PatchSampleF(use_mlp=True, init_type="kaiming", nc=256, n_mlps=3)
    netF.create_mlp([torch.zeros(1, c, 1, 1, 1, device=dev) for c in (128, 256, 128, 64, 32, 16)])
netG.precision = "bf16"
netG = netG.to(dev).train()
netF = netF.to(dev).train()
nopt = Namespace(nce_T=0.33, weigh_rarity=False, balance_denominator=False, weighting_mode="raw")
crits = [SupPatchNCELoss(nopt) for _ in PI.NCE_LAYERS]
optG = torch.optim.AdamW(netG.parameters(), lr=2e-4, weight_decay=1e-5, capturable=True)
optF = torch.optim.AdamW(netF.parameters(), lr=2e-4, weight_decay=1e-5, capturable=True)
A, B, seg = [t.to(dev) for t in PI.step_inputs(S)]
x = torch.cat((A, B))
n_patches = 512 if S >= 128 else 64


def body():
    out, feats = netG(x, list(PI.NCE_LAYERS), False)
    if stage == "unet":
        loss = out.float().square().mean() + sum(f.float().square().mean() for f in feats)
    else:
        pooled, ids = netF(feats, n_patches, None, None, False)
        loss = sum(c(f, seg, i, ft.size()[2:]).mean() for c, f, i, ft in zip(crits, pooled, ids, feats)) / len(crits)
    loss.backward()
    if stage == "full":
        optG.step(); optF.step()
    return loss


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        optG.zero_grad(set_to_none=True); optF.zero_grad(set_to_none=True)
        l0 = body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager loss", float(l0), flush=True)
g = torch.cuda.CUDAGraph()
optG.zero_grad(set_to_none=True); optF.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    loss = body()
torch.cuda.synchronize()
print("captured", flush=True)
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, float(loss), flush=True)
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"replay {1e3 * (time.perf_counter() - t0) / 20:.2f} ms per step", flush=True)
