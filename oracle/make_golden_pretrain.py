"""Generates tests/golden/pretrain_golden.npz by running the REFERENCE's contrastive-pretraining pieces
(/root/reference/pretraining, imported -- never copied) on deterministic synthetic inputs.

Run in the build container only:   python oracle/make_golden_pretrain.py
Holds data only: loss known-answers + gradient probes of SupPatchNCELoss (supcl_model.py:16-226) for its four flag
combinations, outputs of PatchSampleF (pretraining_networks.py:264-519) for given coordinates and seeded MLP
parameters, and one full two-view step record at 64^3 (per-layer losses, total, gradient norms / checksums).
Inputs and parameters are regenerated from seeds by tests (oracle.pretrain_inputs), nothing of the reference travels.

PatchSampleF.forward hard-codes ``.cuda()`` for its masks (pretraining_networks.py:400,405); this script (only)
shims Tensor.cuda / Module.cuda to the identity so the reference runs on the CPU of the build container.
"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/pretraining")
sys.path.insert(0, "/root/reference")

torch.Tensor.cuda = lambda self, *a, **k: self            # noqa: E731  (generator-only shim, see docstring)
torch.nn.Module.cuda = lambda self, *a, **k: self         # noqa: E731

from models.supcl_model import SupPatchNCELoss            # noqa: E402  (the reference itself)
from models.pretraining_networks import PatchSampleF      # noqa: E402
from anatomix.model.network import Unet as RefUnet        # noqa: E402
from oracle import pretrain_inputs as PI                  # noqa: E402
from oracle import unet_ref as R                          # noqa: E402

FLAGS = [(False, False, "raw"), (True, False, "raw"), (False, True, "raw"), (True, True, "sqrt")]


def crit(weigh_rarity, balance, mode):
    return SupPatchNCELoss(Namespace(nce_T=PI.NCE_T, weigh_rarity=weigh_rarity, balance_denominator=balance,
                                     weighting_mode=mode))


def step_record(size, prefix, out):
    """One full two-view step of the REFERENCE modules at size^3 (train-mode BatchNorm, fp32, no autocast: CPU)."""
    kw = R.VARIANTS["anatomix"]
    netG = RefUnet(**kw)
    netG.load_state_dict(R.synthetic_state_dict(kw, 3, gain=2 ** 0.5), strict=True)
    netG.train()
    A, B, seg = PI.step_inputs(size)
    reals = torch.cat((A, B), 0)
    out_seg, feat_kq = netG(reals, PI.NCE_LAYERS, False)
    netF = PatchSampleF(use_mlp=True, init_type="normal", init_gain=0.02, nc=PI.NETF_NC, gpu_ids=[], n_mlps=3)
    netF.create_mlp(feat_kq)
    netF.load_state_dict(PI.mlp_state_dict([f.shape[1] for f in feat_kq], seed=9), strict=True)
    netF.train()
    torch.manual_seed(13)
    pooled, ids = netF(feat_kq, PI.NUM_PATCHES, None, None, False)
    for k, sid in enumerate(ids):
        out[f"{prefix}|ids|{k}"] = sid.numpy().astype(np.int16)
    c = crit(False, False, "raw")
    total = 0.0
    per_layer = []
    for f_kq, sid, feat in zip(pooled, ids, feat_kq):
        l = c(f_kq, seg, sid, feat.size()[2:])
        per_layer.append(l.item())
        total = total + l.mean() * (1.0 / len(PI.NCE_LAYERS))
    total.backward()
    gG = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in netG.parameters() if p.grad is not None))
    gF = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in netF.parameters() if p.grad is not None))
    out[f"{prefix}|per_layer"] = np.array(per_layer)
    out[f"{prefix}|total"] = np.array(total.item())
    out[f"{prefix}|grad_norm_G"] = np.array(gG.item())
    out[f"{prefix}|grad_norm_F"] = np.array(gF.item())
    for name, prm in netG.named_parameters():            # per-parameter gradient norm + a few seeded probes of every layer
        g = prm.grad.double().reshape(-1)
        idx = np.random.RandomState(sum(map(ord, name))).randint(0, g.numel(), min(256, g.numel()))
        out[f"{prefix}|gnorm|{name}"] = np.array(g.norm().item())
        out[f"{prefix}|gidx|{name}"] = idx.astype(np.int64)
        out[f"{prefix}|gval|{name}"] = g[idx].numpy().astype(np.float32)
    out[f"{prefix}|out_probe"] = out_seg.detach().reshape(-1)[:: 9973].numpy().astype(np.float32)
    out[f"{prefix}|bn1_running_mean"] = netG.model[1].running_mean.numpy().copy()
    print(prefix, "per-layer", per_layer, "total", total.item(), "gG", gG.item(), "gF", gF.item())


def main_step128():
    """`python oracle/make_golden_pretrain.py --step128`: the two-view step record at the 128^3 operating size of BASELINE
    configs[2] (per-layer losses, total, gradient norms, per-parameter gradient norms and probes) in its own small file."""
    torch.set_num_threads(8)
    out = {}
    step_record(128, "step128", out)
    path = os.path.join(ROOT, "tests", "golden", "pretrain_step128_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


def main():
    torch.set_num_threads(8)
    out = {}
    # ---- (1) loss known answers + gradients
    for case in PI.LOSS_CASES:
        feats, seg, coords, size = PI.loss_inputs(case)
        for wr, bal, mode in FLAGS:
            f = feats.clone().requires_grad_(True)
            loss = crit(wr, bal, mode)(f, seg, coords, size)
            loss.backward()
            tag = f"loss|{case}|wr{int(wr)}|bal{int(bal)}|{mode}"
            out[tag + "|value"] = np.array(loss.item(), dtype=np.float64)
            idx = np.random.RandomState(7).randint(0, f.numel(), 4096).astype(np.int64)
            out[tag + "|grad_idx"] = idx
            out[tag + "|grad_val"] = f.grad.reshape(-1)[idx].numpy().astype(np.float32)
            out[tag + "|grad_norm"] = np.array(f.grad.double().norm().item())
            print(tag, loss.item())
    # ---- (2) patch sampler + projection head
    # (the reference's given-patch_ids branch leaves `coords` unbound, pretraining_networks.py:432-447,499, so it can
    #  only be driven through its own randperm: the draws are CAPTURED into the fixture, as SURVEY.md section 8c says)
    feats, _ = PI.sampler_inputs()
    netF = PatchSampleF(use_mlp=True, init_type="normal", init_gain=0.02, nc=PI.NETF_NC, gpu_ids=[], n_mlps=3)
    netF.create_mlp(feats)
    netF.load_state_dict(PI.mlp_state_dict([f.shape[1] for f in feats], seed=5), strict=True)
    netF.train()
    torch.manual_seed(11)
    with torch.no_grad():
        pooled, ret_ids = netF(feats, PI.NUM_PATCHES, None, None, False)
    for k, (pf, f) in enumerate(zip(pooled, feats)):
        out[f"sampler|{k}|out"] = pf.numpy().astype(np.float32)
        out[f"sampler|{k}|ids"] = ret_ids[k].numpy().astype(np.int16)
        print("sampler", k, tuple(pf.shape), pf.abs().mean().item())
    # ---- (3) one full two-view step at 64^3 (train-mode BatchNorm, fp32, no autocast: CPU)
    kw = R.VARIANTS["anatomix"]
    netG = RefUnet(**kw)
    netG.load_state_dict(R.synthetic_state_dict(kw, 3, gain=2 ** 0.5), strict=True)
    netG.train()
    A, B, seg = PI.step_inputs(64)
    reals = torch.cat((A, B), 0)
    out_seg, feat_kq = netG(reals, PI.NCE_LAYERS, False)
    netF = PatchSampleF(use_mlp=True, init_type="normal", init_gain=0.02, nc=PI.NETF_NC, gpu_ids=[], n_mlps=3)
    netF.create_mlp(feat_kq)
    netF.load_state_dict(PI.mlp_state_dict([f.shape[1] for f in feat_kq], seed=9), strict=True)
    netF.train()
    torch.manual_seed(13)
    pooled, ids = netF(feat_kq, PI.NUM_PATCHES, None, None, False)
    for k, sid in enumerate(ids):
        out[f"step|ids|{k}"] = sid.numpy().astype(np.int16)
    c = crit(False, False, "raw")
    total = 0.0
    per_layer = []
    for f_kq, sid, feat in zip(pooled, ids, feat_kq):
        l = c(f_kq, seg, sid, feat.size()[2:])
        per_layer.append(l.item())
        total = total + l.mean() * (1.0 / len(PI.NCE_LAYERS))
    total.backward()
    gG = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in netG.parameters() if p.grad is not None))
    gF = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in netF.parameters() if p.grad is not None))
    out["step|per_layer"] = np.array(per_layer)
    out["step|total"] = np.array(total.item())
    out["step|grad_norm_G"] = np.array(gG.item())
    out["step|grad_norm_F"] = np.array(gF.item())
    for name in ("model.0.weight", "model.27.weight", "model.59.weight", "model.65.weight", "model.1.weight", "model.1.bias"):
        g = dict(netG.named_parameters())[name].grad
        out[f"step|grad|{name}"] = np.array([g.double().sum().item(), g.double().norm().item()])
    out["step|out_probe"] = out_seg.detach().reshape(-1)[:: 9973].numpy().astype(np.float32)
    out["step|bn1_running_mean"] = netG.model[1].running_mean.numpy().copy()
    print("step: per-layer", per_layer, "total", total.item(), "gG", gG.item(), "gF", gF.item())
    path = os.path.join(ROOT, "tests", "golden", "pretrain_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    if "--step128" in sys.argv:
        main_step128()
    else:
        main()
