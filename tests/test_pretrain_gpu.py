"""GPU: the HIP supervised-contrastive loss kernel (forward + backward through the C ABI) and the contrastive step."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

import anatomix_amd
from anatomix_amd.pretraining import PatchSampleF, SupPatchNCELoss, contrastive_step
from oracle import pretrain_inputs as PI
from oracle import supcon_ref as S
from oracle import unet_ref as R

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "pretrain_golden.npz"))
FLAGS = [(False, False, "raw"), (True, False, "raw"), (False, True, "raw"), (True, True, "sqrt")]


def _opt(wr, bal, mode):
    return Namespace(nce_T=PI.NCE_T, weigh_rarity=wr, balance_denominator=bal, weighting_mode=mode)


@pytest.mark.parametrize("case", list(PI.LOSS_CASES))
@pytest.mark.parametrize("wr,bal,mode", FLAGS)
def test_hip_loss_forward_backward(device, case, wr, bal, mode):
    feats, seg, coords, size = PI.loss_inputs(case)
    tag = f"loss|{case}|wr{int(wr)}|bal{int(bal)}|{mode}"
    f = feats.to(device).requires_grad_(True)
    crit = SupPatchNCELoss(_opt(wr, bal, mode))
    loss = crit(f, seg.to(device), coords.to(device), size)
    (2.0 * loss).backward()                                    # upstream gradient 2: the Function scales its saved grad
    # vs the reference's own fp32 numbers
    assert abs(loss.item() - float(GOLD[tag + "|value"])) < 2e-5 * abs(loss.item())
    got = 0.5 * f.grad.cpu().reshape(-1)[torch.from_numpy(GOLD[tag + "|grad_idx"])].numpy()
    ref = GOLD[tag + "|grad_val"]
    assert np.abs(got - ref).max() < 5e-5 * np.abs(ref).max()
    # vs the float64 oracle over the whole gradient
    fo = feats.double().requires_grad_(True)
    lo = S.supcon_loss(fo, S.gather_labels(seg, coords, size), PI.NCE_T, wr, bal, mode)
    lo.backward()
    assert abs(loss.item() - lo.item()) < 1e-5 * abs(lo.item())
    err = (0.5 * f.grad.cpu().double() - fo.grad).norm() / fo.grad.norm()
    assert err.item() < 2e-5, err.item()


def test_hip_loss_is_deterministic_and_forward_only_works(device):
    feats, seg, coords, size = PI.loss_inputs("p512c256")
    crit = SupPatchNCELoss(_opt(False, True, "raw"))
    f = feats.to(device).requires_grad_(True)
    l1 = crit(f, seg.to(device), coords.to(device), size)
    l1.backward()
    g1 = f.grad.clone()
    f.grad = None
    l2 = crit(f, seg.to(device), coords.to(device), size)
    l2.backward()
    assert torch.equal(l1, l2) and torch.equal(g1, f.grad)
    with torch.no_grad():
        l3 = crit(feats.to(device), seg.to(device), coords.to(device), size)
    assert torch.equal(l3, l1.detach())


def test_contrastive_step_on_gpu_matches_reference_record(device):
    """Two-view step at 64^3: UNet through torch autograd on the GPU (stock modules), sampling + MLP with torch,
    the six losses on the HIP kernel; against the record captured from the reference on CPU."""
    kw = R.VARIANTS["anatomix"]
    netG = anatomix_amd.Unet(**kw)
    netG.load_state_dict(R.synthetic_state_dict(kw, 3, gain=2 ** 0.5), strict=True)
    netG.allow_torch_path = True
    netG._warned = True
    netG = netG.to(device).train()
    A, B, seg = [t.to(device) for t in PI.step_inputs(64)]
    ids = [torch.from_numpy(GOLD[f"step|ids|{k}"].astype(np.int64)).to(device) for k in range(6)]
    netF = PatchSampleF(use_mlp=True, nc=PI.NETF_NC, n_mlps=3)
    chans = [128, 256, 128, 64, 32, 16]
    netF.create_mlp([torch.zeros(1, c, 1, 1, 1, device=device) for c in chans])
    netF.load_state_dict(PI.mlp_state_dict(chans, seed=9), strict=True)
    netF = netF.to(device).train()
    crits = [SupPatchNCELoss(_opt(False, False, "raw")) for _ in PI.NCE_LAYERS]
    opt_G = torch.optim.AdamW(netG.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    opt_F = torch.optim.AdamW(netF.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    w0 = netG.model[0].weight.detach().clone()
    rec = contrastive_step(netG, netF, crits, A, B, seg, PI.NCE_LAYERS, num_patches=PI.NUM_PATCHES, sample_ids=ids,
                           optimizers=(opt_G, opt_F))
    np.testing.assert_allclose(list(rec["per_layer"].values()), GOLD["step|per_layer"], rtol=2e-3)
    assert abs(rec["loss"] - float(GOLD["step|total"])) < 1e-3 * rec["loss"]
    assert abs(rec["grad_norm_G"] - float(GOLD["step|grad_norm_G"])) < 2e-2 * rec["grad_norm_G"]
    assert abs(rec["grad_norm_F"] - float(GOLD["step|grad_norm_F"])) < 2e-2 * rec["grad_norm_F"]
    assert not torch.equal(w0, netG.model[0].weight.detach())          # AdamW stepped
    assert all(p.grad is None or not p.grad.any() for p in netG.parameters())   # and zero_grad ran
