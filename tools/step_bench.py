"""Contrastive step timing on one GPU: HIP training path vs the stock torch modules (bf16 autocast), 2 x 128^3 views."""
import sys, os, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from argparse import Namespace
import torch
import anatomix_amd
from anatomix_amd.pretraining import PatchSampleF, SupPatchNCELoss, contrastive_step
from oracle import unet_ref as R, pretrain_inputs as PI
dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mode = sys.argv[2] if len(sys.argv) > 2 else "hip"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
kw = R.VARIANTS["anatomix"]
so = sys.stdout; sys.stdout = open(os.devnull, "w")
netG = anatomix_amd.Unet(**kw); netG.load_state_dict(R.synthetic_state_dict(kw, 3, gain=2 ** 0.5)); netG.precision = "bf16"
if mode == "torch":
    netG.allow_torch_path = True; netG._warned = True
    import anatomix_amd.model.train as TR
    TR.unsupported_reason = lambda *a, **k: "forced stock-module run"
netG = netG.to(dev).train()
netF = PatchSampleF(use_mlp=True, nc=256, n_mlps=3)
chans = [128, 256, 128, 64, 32, 16]
netF.create_mlp([torch.zeros(1, c, 1, 1, 1, device=dev) for c in chans]); netF = netF.to(dev).train()
sys.stdout = so
opt = Namespace(nce_T=0.33, weigh_rarity=False, balance_denominator=False, weighting_mode="raw")
crits = [SupPatchNCELoss(opt) for _ in PI.NCE_LAYERS]
oG = torch.optim.AdamW(netG.parameters(), lr=2e-4, weight_decay=1e-5); oF = torch.optim.AdamW(netF.parameters(), lr=2e-4, weight_decay=1e-5)
A, B, seg = [t.to(dev) for t in PI.step_inputs(S)]
def step():
    if mode == "torch":
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return contrastive_step(netG, netF, crits, A, B, seg, PI.NCE_LAYERS, optimizers=(oG, oF))
    return contrastive_step(netG, netF, crits, A, B, seg, PI.NCE_LAYERS, optimizers=(oG, oF))
for _ in range(2): r = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): r = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"{mode} S={S}: {dt*1e3:.1f} ms/step  ({2/dt:.2f} volumes/s through fwd+bwd+opt)  loss {r['loss']:.4f}  gG {r['grad_norm_G']:.3f}  mem {torch.cuda.max_memory_allocated()/1e9:.2f} GB")
