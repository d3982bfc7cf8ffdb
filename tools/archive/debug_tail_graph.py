"""Reproducer: GraphedContrastiveStep with the optimizers in a SECOND graph (tail_graph=True) aborted at its first replay."""
import contextlib, io, sys, os
from argparse import Namespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anatomix_amd
from anatomix_amd.pretraining import GradientBuckets, GraphedContrastiveStep, PatchSampleF, SupPatchNCELoss
from oracle import pretrain_inputs as PI, unet_ref as R
dev = torch.device("cuda:0")
kw = R.VARIANTS["anatomix"]
with contextlib.redirect_stdout(io.StringIO()):
    netG = anatomix_amd.Unet(**kw)
    netG.load_state_dict(R.synthetic_state_dict(kw, 3, gain=2 ** 0.5))
    netF = PatchSampleF(use_mlp=True, init_type="kaiming", nc=256, n_mlps=3)
    netF.create_mlp([torch.zeros(1, c, 1, 1, 1, device=dev) for c in (128, 256, 128, 64, 32, 16)])
netG.precision = "bf16"
netG, netF = netG.to(dev).train(), netF.to(dev).train()
nopt = Namespace(nce_T=0.33, weigh_rarity=False, balance_denominator=False, weighting_mode="raw")
crits = [SupPatchNCELoss(nopt) for _ in PI.NCE_LAYERS]
A, B, seg = [t.to(dev) for t in PI.step_inputs(64)]
opts = (torch.optim.AdamW(netG.parameters(), lr=1e-3, capturable=True), torch.optim.AdamW(netF.parameters(), lr=1e-3, capturable=True))
mode = sys.argv[1] if len(sys.argv) > 1 else "buckets"
buckets = GradientBuckets((netG, netF), bucket_mb=8.0) if mode == "buckets" else None
step = GraphedContrastiveStep(netG, netF, crits, PI.NCE_LAYERS, opts, num_patches=64, warmup=2, grad_buckets=buckets,
                              grad_sync=lambda: None, tail_graph=True)
for i in range(4):
    r = step(A, B, seg)
    torch.cuda.synchronize()
    print(mode, i, r["loss"], r["grad_norm_G"], flush=True)
print("OK", mode)
