"""GPU: the six projection heads and the six losses of a contrastive step as batches (amx_mlp_heads_forward / _backward,
amx_supcon_loss_batch, amx_gather_labels_batch) against the same heads / losses evaluated one by one (amx_mlp_head_*, amx_supcon_loss):
per head the same kernels run in the same order, so outputs, running statistics, losses and every gradient must agree BIT FOR BIT.
(The one-by-one route is itself pinned against the reference's values in tests/test_pretrain_gpu.py.)"""
import copy
from argparse import Namespace

import pytest
import torch

from anatomix_amd.pretraining import PatchSampleF, SupPatchNCELoss
from anatomix_amd.pretraining import mlp_head, supcon

pytestmark = pytest.mark.gpu
CHANS = [128, 256, 128, 64, 32, 16]
DIMS = [(16, 16, 16), (8, 8, 8), (16, 16, 16), (32, 32, 32), (64, 64, 64), (128, 128, 128)]


def _setup(device, views=2, patches=512, seed=0):
    g = torch.Generator().manual_seed(seed)
    netF = PatchSampleF(use_mlp=True, nc=256, n_mlps=3)
    netF.create_mlp([torch.zeros(1, c, 1, 1, 1) for c in CHANS])
    netF = netF.to(device).train()
    rows = [torch.randn(views, patches, c, generator=g).to(device) for c in CHANS]
    coords = [torch.stack([torch.randint(0, d, (patches,), generator=g) for d in dims], 1).to(device) for dims in DIMS]
    seg = torch.randint(0, 5, (1, 1, 128, 128, 128), generator=g).float().to(device)
    return netF, rows, coords, seg


def test_batched_heads_and_losses_equal_the_chains_bit_for_bit(device):
    netF, rows, coords, seg = _setup(device)
    netB = copy.deepcopy(netF)
    opt = Namespace(nce_T=0.33, weigh_rarity=False, balance_denominator=False, weighting_mode="raw")
    crits = [SupPatchNCELoss(opt) for _ in CHANS]
    w = torch.linspace(0.5, 1.5, len(CHANS), device=device)
    # one by one
    ra = [r.clone().requires_grad_(True) for r in rows]
    pooled, _ = netF.forward_rows(ra, coords)
    la = torch.stack([c(f, seg, sid, torch.Size(d)) for c, f, sid, d in zip(crits, pooled, coords, DIMS)])
    (la * w).sum().backward()
    # batched
    rb = [r.clone().requires_grad_(True) for r in rows]
    pooled_b, _ = netB.forward_rows(rb, coords, None, batched=True)
    lb = supcon.batched_losses(crits, pooled_b, seg, coords, DIMS)
    assert lb is not None and lb.shape == (len(CHANS),)
    (lb * w).sum().backward()
    for k in range(len(CHANS)):
        assert torch.equal(pooled[k], pooled_b[k]), k
        assert torch.equal(ra[k].grad, rb[k].grad), k
    assert torch.equal(la, lb)
    pa, pb = dict(netF.named_parameters()), dict(netB.named_parameters())
    assert pa.keys() == pb.keys() and len(pa) > 0
    for name in pa:
        assert pa[name].grad is not None and torch.equal(pa[name].grad, pb[name].grad), name
    ba, bb = dict(netF.named_buffers()), dict(netB.named_buffers())
    for name in ba:
        assert torch.equal(ba[name], bb[name]), name


@pytest.mark.parametrize("rarity,balance,mode", [(True, False, "raw"), (True, True, "sqrt")])
def test_batched_losses_with_weighting_flags(device, rarity, balance, mode):
    netF, rows, coords, seg = _setup(device, seed=3)
    opt = Namespace(nce_T=0.2, weigh_rarity=rarity, balance_denominator=balance, weighting_mode=mode)
    crits = [SupPatchNCELoss(opt) for _ in CHANS]
    feats = [torch.randn(2, 512, 256, device=device).requires_grad_(True) for _ in CHANS]
    feats_b = [f.detach().clone().requires_grad_(True) for f in feats]
    la = torch.stack([c(f, seg, sid, torch.Size(d)) for c, f, sid, d in zip(crits, feats, coords, DIMS)])
    la.sum().backward()
    lb = supcon.batched_losses(crits, feats_b, seg, coords, DIMS)
    lb.sum().backward()
    assert torch.equal(la, lb)
    for a, b in zip(feats, feats_b):
        assert torch.equal(a.grad, b.grad)


def test_a_batch_of_mixed_structure_is_refused(device):
    netF, rows, coords, seg = _setup(device)
    mlps = [getattr(netF, "mlp_%d" % k) for k in range(2)]
    xs = [rows[0].flatten(0, 1), rows[1].flatten(0, 1)[:512]]          # different row counts
    assert mlp_head.batchable(mlps, xs) is not None
    with pytest.raises(RuntimeError, match="batch"):
        mlp_head.run_heads(mlps, xs)
    opt = Namespace(nce_T=0.33, weigh_rarity=False, balance_denominator=False, weighting_mode="raw")
    opt2 = Namespace(nce_T=0.5, weigh_rarity=False, balance_denominator=False, weighting_mode="raw")
    f = [torch.randn(2, 512, 256, device=device) for _ in range(2)]
    assert supcon.batched_losses([SupPatchNCELoss(opt), SupPatchNCELoss(opt2)], f, seg, coords[:2], DIMS[:2]) is None
