"""Where does the contrastive step's wall time go: host enqueue time vs GPU busy time."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from argparse import Namespace
import torch
import anatomix_amd
from anatomix_amd.pretraining import PatchSampleF, SupPatchNCELoss, contrastive_step
import anatomix_amd.pretraining.step as STEP
from oracle import unet_ref as R, pretrain_inputs as PI
dev = torch.device("cuda:0")
kw = R.VARIANTS["anatomix"]
so = sys.stdout; sys.stdout = open(os.devnull, "w")
netG = anatomix_amd.Unet(**kw); netG.load_state_dict(R.synthetic_state_dict(kw, 3, gain=2 ** 0.5)); netG.precision = "bf16"
netG = netG.to(dev).train()
netF = PatchSampleF(use_mlp=True, nc=256, n_mlps=3)
netF.create_mlp([torch.zeros(1, c, 1, 1, 1, device=dev) for c in (128, 256, 128, 64, 32, 16)]); netF = netF.to(dev).train()
sys.stdout = so
opt = Namespace(nce_T=0.33, weigh_rarity=False, balance_denominator=False, weighting_mode="raw")
crits = [SupPatchNCELoss(opt) for _ in PI.NCE_LAYERS]
oG = torch.optim.AdamW(netG.parameters(), lr=2e-4); oF = torch.optim.AdamW(netF.parameters(), lr=2e-4)
A, B, seg = [t.to(dev) for t in PI.step_inputs(128)]
for _ in range(3): contrastive_step(netG, netF, crits, A, B, seg, PI.NCE_LAYERS, optimizers=(oG, oF))
# host-only time: forward of netG (enqueue) without syncing
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(5): contrastive_step(netG, netF, crits, A, B, seg, PI.NCE_LAYERS, optimizers=(oG, oF))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
pr.disable()
print(f"wall {dt*1e3:.2f} ms/step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
