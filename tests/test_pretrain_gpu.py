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


@pytest.mark.parametrize("seg_shape,fsize", [((128, 128, 128), (64, 64, 64)), ((128, 128, 128), (8, 8, 8)), ((40, 52, 36), (13, 20, 36)),
                                             ((30, 30, 30), (30, 30, 30)), ((17, 9, 33), (5, 4, 7))])
def test_label_gather_kernel_equals_interpolate_and_index(device, seg_shape, fsize):
    """amx_gather_labels (the criterion's one-launch label path) against F.interpolate(seg, size, 'nearest')[coords] -> round -> int ->
    repeat, on divisible and non-divisible size ratios."""
    import ctypes
    from anatomix_amd import _lib
    g = torch.Generator().manual_seed(4)
    seg = torch.randint(0, 9, (1, 1) + seg_shape, generator=g).float().to(device)
    P = 300
    coords = torch.stack([torch.randint(0, n, (P,), generator=g) for n in fsize], dim=1).to(device)
    want = SupPatchNCELoss.gather_labels(seg, coords, fsize)[0].round().to(torch.int32).repeat(2)
    lib = _lib.load()
    got = torch.empty(2 * P, dtype=torch.int32, device=device)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _lib.check(lib.amx_gather_labels(_lib.ptr(seg), *seg_shape, _lib.ptr(coords), P, *fsize, 2, _lib.ptr(got), st))
    assert torch.equal(got, want)


def test_criterion_with_host_coordinates_takes_the_indexing_route(device):
    """The one-launch label path hands raw pointers to a kernel: coordinates on the host (what the old indexing route accepted) must
    not reach it.  Same loss either way."""
    feats, seg, coords, size = PI.loss_inputs("p512c256")
    crit = SupPatchNCELoss(_opt(False, True, "raw"))
    f = feats.to(device)
    on_device = crit(f, seg.to(device), coords.to(device), size)
    on_host = crit(f, seg.to(device), coords, size)                       # CPU int64 [P, 3]
    assert torch.equal(on_device, on_host)
    with pytest.raises((IndexError, RuntimeError)):
        crit(f, seg.to(device), coords[:100].to(device), size)            # fewer rows than patches: a Python error, never an out-of-bounds read


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


# ---- projection head on the HIP kernels (csrc/amx_mlp.hip) -------------------------------------------------------------

def _head(cin, width, n_mlps, act, seed):
    import torch.nn as nn
    torch.manual_seed(seed)
    A = (lambda: nn.ReLU(inplace=True)) if act == "relu" else (lambda: nn.LeakyReLU(0.3, inplace=True))
    mods = [nn.Linear(cin, width, bias=False), nn.BatchNorm1d(width), A()]
    for _ in range(n_mlps - 2):
        mods += [nn.Linear(width, width, bias=False), nn.BatchNorm1d(width), A()]
    mods += [nn.Linear(width, width, bias=False), nn.BatchNorm1d(width, affine=False)]
    m = nn.Sequential(*mods)
    for mod in m:
        if isinstance(mod, nn.BatchNorm1d) and mod.weight is not None:
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.2)
    return m


@pytest.mark.gpu
def test_lazy_step_records_hold_the_numbers_of_the_blocking_ones():
    """GraphedContrastiveStep(lazy_scalars=True) returns at once and reads the scalars on first use: twelve replays enqueued without a
    host synchronisation in between (more than the ring of pinned slots) must report exactly what the blocking mode reports for the
    same seed, step by step."""
    import contextlib, io
    from argparse import Namespace
    import anatomix_amd
    from anatomix_amd.pretraining import GraphedContrastiveStep, PatchSampleF, SupPatchNCELoss, FusedAdamW
    from anatomix_amd.pretraining.step import StepRecord
    from oracle import pretrain_inputs as PI, unet_ref as R
    dev = torch.device("cuda:0")
    kw = R.VARIANTS["anatomix"]
    nopt = Namespace(nce_T=0.33, weigh_rarity=False, balance_denominator=False, weighting_mode="raw")
    A, B, seg = [t.to(dev) for t in PI.step_inputs(64)]

    def run(lazy):
        torch.manual_seed(11)
        with contextlib.redirect_stdout(io.StringIO()):
            netG = anatomix_amd.Unet(**kw)
            netG.load_state_dict(R.synthetic_state_dict(kw, 3, gain=2 ** 0.5))
            netF = PatchSampleF(use_mlp=True, init_type="kaiming", nc=256, n_mlps=3)
            netF.create_mlp([torch.zeros(1, c, 1, 1, 1, device=dev) for c in (128, 256, 128, 64, 32, 16)])
        netG.precision = "bf16"
        netG, netF = netG.to(dev).train(), netF.to(dev).train()
        crits = [SupPatchNCELoss(nopt) for _ in PI.NCE_LAYERS]
        opts = (FusedAdamW(netG.parameters(), lr=1e-3), FusedAdamW(netF.parameters(), lr=1e-3))
        step = GraphedContrastiveStep(netG, netF, crits, PI.NCE_LAYERS, opts, num_patches=64, warmup=2, lazy_scalars=lazy)
        torch.manual_seed(12)
        recs = [step(A, B, seg) for _ in range(12)]
        return [(r["loss"], r["grad_norm_G"], r["grad_norm_F"], tuple(r["per_layer"].values())) for r in recs], recs

    want, _ = run(False)
    got, recs = run(True)
    assert all(isinstance(r, StepRecord) for r in recs) and set(recs[0]) == {"loss", "per_layer", "grad_norm_G", "grad_norm_F", "sample_ids", "out"}
    assert got == want
    assert all(np.isfinite(v[0]) for v in got) and len({v[0] for v in got}) > 1


@pytest.mark.parametrize("n,cin,width,n_mlps,act", [(1024, 16, 256, 3, "relu"), (1024, 256, 256, 3, "relu"),
                                                    (128, 128, 256, 2, "relu"), (1000, 64, 128, 3, "lrelu"),
                                                    (2048, 32, 64, 2, "lrelu"), (7, 4, 8, 3, "relu")])
def test_mlp_head_matches_float64_modules(n, cin, width, n_mlps, act):
    import copy
    from anatomix_amd.pretraining import mlp_head
    dev = torch.device("cuda:0")
    ref = _head(cin, width, n_mlps, act, seed=n + cin).double().train()
    hip = copy.deepcopy(ref).float().to(dev).train()
    torch.manual_seed(1)
    x64 = torch.randn(n, cin, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(n, width, dtype=torch.float64)
    y64 = ref(x64)
    (y64 * gy).sum().backward()
    x = x64.detach().float().to(dev).requires_grad_(True)
    assert mlp_head.unsupported_reason(hip, x) is None
    y = mlp_head.run_head(hip, x)
    (y * gy.float().to(dev)).sum().backward()

    def rel(a, b):
        return (a.double().cpu() - b).abs().max().item() / (b.abs().max().item() + 1e-30)

    assert rel(y, y64.detach()) < 2e-5
    assert rel(x.grad, x64.grad) < 2e-4
    for (name, p), (_, q) in zip(hip.named_parameters(), ref.named_parameters()):
        assert rel(p.grad, q.grad) < 2e-4, name
    for (name, b), (_, c) in zip(hip.named_buffers(), ref.named_buffers()):
        if "num_batches" in name:
            assert int(b) == int(c) == 1
        else:
            assert rel(b, c) < 1e-5, name


def test_mlp_head_is_deterministic_and_used_by_the_sampler():
    from anatomix_amd.pretraining import PatchSampleF, mlp_head
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    netF = PatchSampleF(use_mlp=True, init_type="kaiming", nc=256, n_mlps=3)
    feats = [torch.randn(2, c, s, s, s, device=dev, requires_grad=True) for c, s in ((128, 8), (16, 32))]
    netF.create_mlp(feats)
    netF = netF.to(dev).train()
    calls = []
    orig = mlp_head.run_head
    mlp_head.run_head = lambda m, x: calls.append(x.shape) or orig(m, x)
    try:
        out, ids = netF(feats, 512, None, None)
        out2, _ = netF([f.detach() for f in feats], 512, ids, None)
    finally:
        mlp_head.run_head = orig
    assert len(calls) == 4 and out[0].shape == (2, 512, 256)
    assert all(torch.equal(a, b) for a, b in zip(out, out2))            # same ids, same batch -> bit-identical
    sum(o.square().mean() for o in out).backward()
    assert all(f.grad is not None and torch.isfinite(f.grad).all() for f in feats)
    assert all(p.grad is not None for p in netF.parameters())
    netF.eval()                                                          # eval mode: running statistics, stock modules
    calls.clear()
    mlp_head.run_head = lambda m, x: calls.append(x.shape) or orig(m, x)
    try:
        with torch.no_grad():
            netF([f.detach() for f in feats], 512, ids, None)
    finally:
        mlp_head.run_head = orig
    assert not calls


def test_distinct_coordinate_sampler():
    """amx_sample_coords behind PatchSampleF's no-mask branch: distinct in-range coordinates, reproducible from torch's
    generator, uniform over the grid, and robust when the draws collide a lot."""
    from anatomix_amd.pretraining import PatchSampleF
    dev = torch.device("cuda:0")
    # (the last two: the largest draw count the entry takes, and a grid of more than 2^31 voxels -- 64-bit keys in the kernel's hash table)
    for dims, num in (((128, 128, 128), 512), ((16, 16, 16), 512), ((20, 12, 18), 300), ((64, 48), 256), ((128, 128, 128), 2048),
                      ((2048, 2048, 1024), 2048)):
        nvox = int(np.prod(dims))
        torch.manual_seed(5)
        c1 = PatchSampleF._sample_distinct(dev, nvox, num, list(dims))
        torch.manual_seed(5)
        c2 = PatchSampleF._sample_distinct(dev, nvox, num, list(dims))
        assert c1.shape == (num, len(dims)) and c1.dtype == torch.int64 and torch.equal(c1, c2)
        c = c1.cpu().numpy()
        assert (c >= 0).all() and (c < np.array(dims)).all()
        flat = np.ravel_multi_index(tuple(c.T), dims)
        assert len(np.unique(flat)) == num
        # the kept values are the first distinct draws, in draw order
        torch.manual_seed(5)
        draws = torch.randint(nvox, (2 * num,), device=dev, dtype=torch.int64).cpu().numpy()
        _, first = np.unique(draws, return_index=True)
        assert (flat == draws[np.sort(first)][:num]).all()
    # uniformity: pooled over many calls every octant of the grid gets its share
    torch.manual_seed(0)
    allc = torch.cat([PatchSampleF._sample_distinct(dev, 64 ** 3, 512, [64, 64, 64]) for _ in range(40)]).cpu().numpy()
    octant = (allc >= 32) @ np.array([4, 2, 1])
    counts = np.bincount(octant, minlength=8) / len(allc)
    assert np.abs(counts - 0.125).max() < 0.02
    # small grids (fewer than 8 x num voxels, <= 4096): a uniformly random permutation prefix from one key per voxel (amx_sample_perm)
    for dims, num in (((8, 8, 8), 512), ((8, 8, 8), 100), ((16, 16, 16), 600), ((5, 7, 3), 105), ((12, 20), 64)):
        torch.manual_seed(9)
        p1 = PatchSampleF._sample_perm(dev, num, list(dims))
        torch.manual_seed(9)
        p2 = PatchSampleF._sample_perm(dev, num, list(dims))
        assert p1.shape == (num, len(dims)) and p1.dtype == torch.int64 and torch.equal(p1, p2)
        c = p1.cpu().numpy()
        assert (c >= 0).all() and (c < np.array(dims)).all()
        flat = np.ravel_multi_index(tuple(c.T), dims)
        assert len(np.unique(flat)) == num
        torch.manual_seed(9)                                             # the order is the order of the keys, ties by index
        keys = torch.randint(1 << 62, (int(np.prod(dims)),), device=dev, dtype=torch.int64).cpu().numpy() & ((1 << 50) - 1)
        assert (flat == np.lexsort((np.arange(len(keys)), keys))[:num]).all()
    torch.manual_seed(1)                                                 # every voxel is equally likely to come first
    firsts = torch.cat([PatchSampleF._sample_perm(dev, 1, [4, 4, 4]) for _ in range(640)]).cpu().numpy()
    counts = np.bincount(np.ravel_multi_index(tuple(firsts.T), (4, 4, 4)), minlength=64)
    assert counts.min() >= 1 and counts.max() <= 30                      # mean 10
    netF = PatchSampleF(use_mlp=False)
    feats = [torch.randn(2, 4, 32, 32, 32, device=dev), torch.randn(2, 4, 8, 8, 8, device=dev)]
    out, ids = netF(feats, 512, None, None)
    assert out[0].shape == (2 * 512, 4) and ids[0].shape == (512, 3) and ids[1].shape == (512, 3)   # no MLP: [views * P, C]
    f0 = feats[0][:, :, ids[0][:, 0], ids[0][:, 1], ids[0][:, 2]].permute(0, 2, 1).flatten(0, 1)
    assert torch.equal(out[0], f0)
    assert len(torch.unique(ids[1][:, 0] * 64 + ids[1][:, 1] * 8 + ids[1][:, 2])) == 512


def test_graphed_contrastive_step_matches_eager_on_the_same_coordinates():
    """GraphedContrastiveStep: forward + sampling + heads + losses + backward + gradient norms replayed from one HIP graph.
    Every replay draws new coordinates; re-running the eager step on the coordinates a replay used must give its numbers."""
    import contextlib, io
    from argparse import Namespace
    import anatomix_amd
    from anatomix_amd.pretraining import GraphedContrastiveStep, PatchSampleF, SupPatchNCELoss, contrastive_step
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
    graphed = GraphedContrastiveStep(netG, netF, crits, PI.NCE_LAYERS, None, num_patches=64, warmup=2)
    r1 = graphed(A, B, seg)
    ids1 = [i.clone() for i in r1["sample_ids"]]
    grads1 = [p.grad.clone() for p in netG.parameters()]
    # (the eager reference runs on copies: the graph owns the .grad tensors of the captured modules -- resetting them to None
    # would detach the modules from the memory the replays write)
    import copy
    netG2, netF2 = copy.deepcopy(netG), copy.deepcopy(netF)
    for p in list(netG2.parameters()) + list(netF2.parameters()):
        p.grad = None
    ref = contrastive_step(netG2, netF2, crits, A, B, seg, PI.NCE_LAYERS, num_patches=64, optimizers=None, sample_ids=ids1)
    assert abs(r1["loss"] - ref["loss"]) < 1e-5 * abs(ref["loss"]) and abs(r1["grad_norm_G"] - ref["grad_norm_G"]) < 1e-4 * ref["grad_norm_G"]
    for g, p in zip(grads1, netG2.parameters()):
        assert torch.equal(g, p.grad)                       # same kernels, same inputs: bit-identical gradients
    r2 = graphed(A, B, seg)                                 # second replay: new coordinates, same weights -> a different loss
    assert not all(torch.equal(a, b) for a, b in zip(ids1, r2["sample_ids"])) and r2["loss"] != r1["loss"]
    # with capturable optimizers the update is part of the graph and the loss goes down over replays
    opts = (torch.optim.AdamW(netG.parameters(), lr=1e-3, capturable=True), torch.optim.AdamW(netF.parameters(), lr=1e-3, capturable=True))
    trained = GraphedContrastiveStep(netG, netF, crits, PI.NCE_LAYERS, opts, num_patches=64, warmup=2)
    assert trained.opt_in_graph
    losses = [trained(A, B, seg)["loss"] for _ in range(12)]
    assert all(np.isfinite(losses)) and np.mean(losses[-3:]) < np.mean(losses[:3]), losses


def test_graphed_step_with_gradient_sync_hook_runs_the_optimizers_eagerly():
    """Data-parallel shape of GraphedContrastiveStep: forward + backward in the graph, then grad_sync (the all-reduce), the
    gradient norms and ordinary (non-capturable) AdamW outside it."""
    import contextlib, io
    from argparse import Namespace
    import anatomix_amd
    from anatomix_amd.pretraining import GraphedContrastiveStep, PatchSampleF, SupPatchNCELoss
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
    opts = (torch.optim.AdamW(netG.parameters(), lr=1e-3), torch.optim.AdamW(netF.parameters(), lr=1e-3))
    calls = []

    def grad_sync():                                         # stands in for the all-reduce: halve, as averaging two equal ranks' sums would
        calls.append(1)
        for p in netG.parameters():
            p.grad.mul_(1.0)

    step = GraphedContrastiveStep(netG, netF, crits, PI.NCE_LAYERS, opts, num_patches=64, grad_sync=grad_sync, warmup=2)
    assert not step.opt_in_graph
    w0 = next(netG.parameters()).detach().clone()
    losses = [step(A, B, seg)["loss"] for _ in range(10)]
    assert len(calls) == 2 + 10                              # warm-up steps + one per call
    assert all(np.isfinite(losses)) and np.mean(losses[-3:]) < np.mean(losses[:3]), losses
    assert not torch.equal(w0, next(netG.parameters()).detach())


@pytest.mark.parametrize("optimizer", ["torch", "fused"])
def test_graphed_step_with_gradient_buckets(optimizer):
    """The data-parallel shape bench.py runs for N > 1, exercised on one GPU: gradients are views into GradientBuckets' flat
    buffers, [graph 1: forward + backward + gather of the gradients into the buckets] -> bucket sync (the RCCL all-reduce; a
    no-op on one rank, forced through the same code path here) -> [graph 2: gradient norms + capturable AdamW]."""
    import contextlib, io
    from argparse import Namespace
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
    if optimizer == "fused":                                   # what bench.py builds: one amx_adamw_step launch per optimizer, reading the views
        from anatomix_amd.pretraining import FusedAdamW
        opts = (FusedAdamW(netG.parameters(), lr=1e-3), FusedAdamW(netF.parameters(), lr=1e-3))
    else:
        opts = (torch.optim.AdamW(netG.parameters(), lr=1e-3, capturable=True), torch.optim.AdamW(netF.parameters(), lr=1e-3, capturable=True))
    buckets = GradientBuckets((netG, netF), bucket_mb=8.0)
    assert 2 <= len(buckets.buckets) <= 8 and buckets.nbytes == 4 * sum(p.numel() for n in (netG, netF) for p in n.parameters())
    calls = []
    step = GraphedContrastiveStep(netG, netF, crits, PI.NCE_LAYERS, opts, num_patches=64, warmup=2, grad_buckets=buckets,
                                  grad_sync=lambda: calls.append(1))
    assert not step.opt_in_graph and step.tail_in_own_graph
    w0 = next(netG.parameters()).detach().clone()
    recs = [step(A, B, seg) for _ in range(10)]
    assert step.tail_graph is not None
    losses = [r["loss"] for r in recs]
    assert len(calls) == 2 + 10
    flat_ptrs = {v.data_ptr() for vs in buckets.views for v in vs}
    assert all(p.grad.data_ptr() in flat_ptrs for p in netG.parameters())      # the optimizer reads the bucket views
    assert all(np.isfinite(losses)) and all(r["grad_norm_G"] > 0 for r in recs)
    assert np.mean(losses[-3:]) < np.mean(losses[:3]), losses
    assert not torch.equal(w0, next(netG.parameters()).detach())
