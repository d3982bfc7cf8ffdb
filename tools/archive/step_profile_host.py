"""Where the HOST time of the contrastive step goes (cProfile over a few steps) next to wall time per step."""
import cProfile, io, os, pstats, sys, time
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import anatomix_amd
from anatomix_amd.pretraining import PatchSampleF, SupPatchNCELoss, contrastive_step
from oracle import unet_ref as R, pretrain_inputs as PI

dev = torch.device("cuda:0")
kw = R.VARIANTS["anatomix"]
so = sys.stdout; sys.stdout = open(os.devnull, "w")
netG = anatomix_amd.Unet(**kw); netG.load_state_dict(R.synthetic_state_dict(kw, 3, gain=2 ** 0.5)); netG.precision = "bf16"
netG = netG.to(dev).train()
netF = PatchSampleF(use_mlp=True, init_type="kaiming", nc=256, n_mlps=3)
netF.create_mlp([torch.zeros(1, c, 1, 1, 1, device=dev) for c in (128, 256, 128, 64, 32, 16)])
sys.stdout = so
netF = netF.to(dev).train()
nopt = Namespace(nce_T=0.33, weigh_rarity=False, balance_denominator=False, weighting_mode="raw")
crits = [SupPatchNCELoss(nopt) for _ in PI.NCE_LAYERS]
opts = (torch.optim.AdamW(netG.parameters(), lr=2e-4, weight_decay=1e-5), torch.optim.AdamW(netF.parameters(), lr=2e-4, weight_decay=1e-5))
A, B, seg = [t.to(dev) for t in PI.step_inputs(128)]
step = lambda: contrastive_step(netG, netF, crits, A, B, seg, PI.NCE_LAYERS, num_patches=512, optimizers=opts)
for _ in range(5):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize(); print(f"wall per step {1e3 * (time.perf_counter() - t0) / 20:.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
