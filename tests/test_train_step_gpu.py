"""GPU: the differentiable train-mode forward of Unet on the HIP kernels (anatomix_amd/model/train.py) against the stock
torch modules (fp32 autograd on the same device) and against the step record captured from the reference."""
import copy
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from _util import rel_l2
import anatomix_amd
from anatomix_amd.pretraining import PatchSampleF, SupPatchNCELoss, contrastive_step
from oracle import pretrain_inputs as PI
from oracle import unet_ref as R

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "pretrain_golden.npz"))
KW = R.VARIANTS["anatomix"]


def _pair(device, precision="bf16"):
    """The same network twice: `hip` runs the HIP training path, `ref` the stock modules (fp32)."""
    hip = anatomix_amd.Unet(**KW)
    hip.load_state_dict(R.synthetic_state_dict(KW, 3, gain=2 ** 0.5), strict=True)
    hip.precision = precision
    ref = copy.deepcopy(hip)
    ref.allow_torch_path, ref._warned = True, True
    return hip.to(device).train(), ref.to(device).train()


def _ref_forward(ref, x, layers):
    return ref._forward_torch(x, layers, False, False)


def _pair_kw(device, kw, precision, seed=3, gain=1.0):
    hip = anatomix_amd.Unet(**kw)
    hip.load_state_dict(R.synthetic_state_dict(kw, seed, gain=gain), strict=True)
    hip.precision = precision
    ref = copy.deepcopy(hip)
    ref.allow_torch_path, ref._warned = True, True
    return hip.to(device).train(), ref.to(device).train()


def _compare(hip, ref, x, layers, device, loss_scale=1.0):
    out_h, feats_h = hip(x, layers)
    out_r, feats_r = _ref_forward(ref, x, layers)
    errs = {"out": rel_l2(out_h.detach().cpu(), out_r.detach().cpu())}
    for l, a, b in zip(sorted(layers), feats_h, feats_r):
        assert a.shape == b.shape, l
        errs[f"tap{l}"] = rel_l2(a.detach().cpu(), b.detach().cpu())
    g = torch.Generator().manual_seed(5)
    cots = [torch.randn(f.shape, generator=g).to(device) / f[0].numel() ** 0.5 for f in feats_r]
    loss_h = sum((f * c).sum() for f, c in zip(feats_h, cots)) + 0.1 * out_h.square().mean()
    loss_r = sum((f * c).sum() for f, c in zip(feats_r, cots)) + 0.1 * out_r.square().mean()
    (loss_h * loss_scale).backward()       # f16 gradients need the usual loss scaling (the reference runs a GradScaler)
    loss_r.backward()
    for ph in hip.parameters():
        ph.grad /= loss_scale
    gerrs, cos = {}, {}
    for (name, ph), (_, pr) in zip(hip.named_parameters(), ref.named_parameters()):
        assert ph.grad is not None and ph.grad.shape == pr.grad.shape, name
        gerrs[name] = rel_l2(ph.grad.cpu(), pr.grad.cpu())
        a, b = ph.grad.double().flatten().cpu(), pr.grad.double().flatten().cpu()
        cos[name] = (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()
    return errs, gerrs, cos


@pytest.mark.parametrize("kw,layers", [
    (dict(dimension=3, input_nc=1, output_nc=16, num_downs=1, ngf=16), [0, 3, 5, 10, 13, 17, 20]),
    (dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=16, doubleconv=False, activation="lrelu"), [0, 3, 7, 11, 15, 19]),
    (dict(dimension=3, input_nc=1, output_nc=32, num_downs=2, ngf=32), [3, 13, 20, 27, 34]),
])
def test_shallow_networks_forward_and_gradients_match_stock_modules(device, kw, layers):
    """Short networks keep the rounding amplification of train-mode BatchNorm small, so f16 storage must agree with the
    fp32 stock modules tightly: this is the bug-catching end-to-end check (a mis-routed gradient is an O(1) error)."""
    hip, ref = _pair_kw(device, kw, "f16")
    x = R.synthetic_input(11, 2, (32, 32, 64)).to(device)
    errs, gerrs, cos = _compare(hip, ref, x, layers, device, loss_scale=4096.0)
    print("fwd", {k: f"{v:.2e}" for k, v in errs.items()})
    print("grad worst", max(gerrs.values()), "min cos", min(cos.values()))
    assert max(errs.values()) < 1e-2, errs
    # The activations of the two runs differ by ~2e-3, which flips the ReLU mask of the voxels whose pre-activation lies
    # within that distance of zero: a fraction f ~ 2e-3 of the mask, i.e. a sqrt(2f) ~ 5e-2 relative change of the
    # gradient that is a property of comparing two nearby forwards, not of the backward kernels (those are exact to
    # rounding: tests/test_train_ops_gpu.py).  The LAST layers, upstream of no mask, must agree tightly.
    last = [k for k in gerrs if k.startswith(f"model.{len(hip.model) - 1}.")]
    assert all(gerrs[k] < 2e-3 for k in last), {k: gerrs[k] for k in last}
    assert max(gerrs.values()) < 0.2 and min(cos.values()) > 0.98, (max(gerrs.values()), min(cos.values()))
    for i, m in enumerate(hip.model):
        if isinstance(m, torch.nn.BatchNorm3d):
            torch.testing.assert_close(m.running_mean, ref.model[i].running_mean, rtol=2e-3, atol=2e-4)
            torch.testing.assert_close(m.running_var, ref.model[i].running_var, rtol=5e-3, atol=2e-4)
            assert int(m.num_batches_tracked) == 1


def _autocast_reference(ref, x, layers, device):
    """The reference's own execution mode: the stock modules under torch.autocast(bf16) (supcl_model.py:623-626), against which
    the fp32 stock modules are the same yardstick."""
    auto = copy.deepcopy(ref)
    auto.allow_torch_path, auto._warned = True, True
    for p_ in auto.parameters():
        p_.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out, feats = _ref_forward(auto, x, layers)
    return auto, out.float(), [f.float() for f in feats]


@pytest.mark.parametrize("precision,fwd_tol,cos_min", [("f16", 8e-2, 0.75), ("bf16", 0.5, None)])
def test_full_6m_network_train_step_sanity(device, precision, fwd_tol, cos_min):
    """20 train-mode conv+BatchNorm layers on random weights amplify storage rounding by ~1.28x per layer (measured:
    f16 3e-4 -> 4e-2, bf16 3e-3 -> 0.27 at the output; the reference's own bf16 autocast sits at the bf16 level).
    f16 storage is held to absolute bounds.  bf16 storage -- the mode the step actually runs in -- is held to the reference's
    OWN execution mode: the stock modules under torch.autocast(bf16) deviate from the fp32 stock modules through the same
    mechanism, and the HIP path must not be further from fp32 than that (forward taps, gradient directions of the last and the
    shallow layers, and on average over all parameters)."""
    hip, ref = _pair(device, precision)
    A, B, _ = PI.step_inputs(64)
    x = torch.cat((A, B)).to(device)
    layers = [0, 7, 27, 31, 38, 45, 52, 65]
    errs, gerrs, cos = _compare(hip, ref, x, layers, device, loss_scale=4096.0 if precision == "f16" else 1.0)
    print(precision, "fwd", {k: f"{v:.2e}" for k, v in errs.items()})
    print(precision, "grad worst", max(gerrs.values()), "min cos", min(cos.values()))
    assert errs["tap0"] < (1e-3 if precision == "f16" else 6e-3)
    assert max(errs.values()) < fwd_tol, errs
    if cos_min is not None:
        assert min(cos.values()) > cos_min, {k: v for k, v in cos.items() if v < 0.99}
        return
    # ---- bf16: side by side with the reference's bf16 autocast (same cotangents as _compare)
    auto, out_a, feats_a = _autocast_reference(ref, x, layers, device)
    out_r, feats_r = _ref_forward(ref, x, layers)
    g = torch.Generator().manual_seed(5)
    cots = [torch.randn(f.shape, generator=g).to(device) / f[0].numel() ** 0.5 for f in feats_r]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out2, feats2 = _ref_forward(auto, x, layers)
        loss_a = sum((f.float() * c).sum() for f, c in zip(feats2, cots)) + 0.1 * out2.float().square().mean()
    loss_a.backward()
    errs_a = {"out": rel_l2(out_a.detach().cpu(), out_r.detach().cpu())}
    cos_a = {}
    for (name, pa), (_, pr) in zip(auto.named_parameters(), ref.named_parameters()):
        a, b = pa.grad.double().flatten().cpu(), pr.grad.double().flatten().cpu()
        cos_a[name] = (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()
    mean = lambda d: float(np.mean(list(d.values())))
    key = ["model.65.weight", "model.62.weight", "model.59.weight", "model.0.weight", "model.3.weight"]
    print("bf16 autocast (stock modules) vs fp32: out", errs_a["out"], "mean cos", mean(cos_a), {k: round(cos_a[k], 3) for k in key})
    print("bf16 HIP                      vs fp32: out", errs["out"], "mean cos", mean(cos), {k: round(cos[k], 3) for k in key})
    assert errs["out"] <= 1.5 * errs_a["out"] + 1e-2
    assert mean(cos) >= mean(cos_a) - 0.1
    for k in key:
        assert cos[k] >= cos_a[k] - 0.15, (k, cos[k], cos_a[k])


def test_contrastive_step_fully_on_hip_matches_reference_record(device):
    """netG forward + backward on the HIP kernels (bf16 storage, as the reference's bf16 autocast), losses on the HIP kernel."""
    hip, _ = _pair(device, "bf16")
    A, B, seg = [t.to(device) for t in PI.step_inputs(64)]
    ids = [torch.from_numpy(GOLD[f"step|ids|{k}"].astype(np.int64)).to(device) for k in range(6)]
    netF = PatchSampleF(use_mlp=True, nc=PI.NETF_NC, n_mlps=3)
    chans = [128, 256, 128, 64, 32, 16]
    netF.create_mlp([torch.zeros(1, c, 1, 1, 1, device=device) for c in chans])
    netF.load_state_dict(PI.mlp_state_dict(chans, seed=9), strict=True)
    netF = netF.to(device).train()
    opt = Namespace(nce_T=PI.NCE_T, weigh_rarity=False, balance_denominator=False, weighting_mode="raw")
    crits = [SupPatchNCELoss(opt) for _ in PI.NCE_LAYERS]
    opt_G = torch.optim.AdamW(hip.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    opt_F = torch.optim.AdamW(netF.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    rec = contrastive_step(hip, netF, crits, A, B, seg, PI.NCE_LAYERS, num_patches=PI.NUM_PATCHES, sample_ids=ids,
                           optimizers=(opt_G, opt_F))
    got = np.array(list(rec["per_layer"].values()))
    print("per-layer", got, "ref", GOLD["step|per_layer"], "gG", rec["grad_norm_G"], float(GOLD["step|grad_norm_G"]))
    # measured: losses 1e-4, gradient norm of the UNet 0.5 %, of the heads 0.1 % (bf16 storage against the fp32 record)
    np.testing.assert_allclose(got, GOLD["step|per_layer"], rtol=1e-3)
    assert abs(rec["loss"] - float(GOLD["step|total"])) < 1e-3 * rec["loss"]
    assert abs(rec["grad_norm_G"] - float(GOLD["step|grad_norm_G"])) < 0.02 * rec["grad_norm_G"]
    assert abs(rec["grad_norm_F"] - float(GOLD["step|grad_norm_F"])) < 0.02 * rec["grad_norm_F"]


def test_segmentation_finetuning_composition_trains(device):
    """The finetuning caller (anatomix/segmentation/segmentation_utils.py:93-116): nn.Sequential(Unet, 1x1x1 conv head),
    .train(), cross-entropy, optimizer steps -- runs on the HIP training path without opting into the stock modules and
    the loss goes down."""
    torch.manual_seed(0)
    net = anatomix_amd.Unet(**KW)
    net.load_state_dict(R.synthetic_state_dict(KW, 1, gain=2 ** 0.5), strict=True)
    net.precision = "bf16"
    head = torch.nn.Conv3d(16, 4, kernel_size=1)                      # monai's UnetOutBlock(3, feat, n_classes + 1) is this conv
    model = torch.nn.Sequential(net, head).to(device).train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3)
    x = R.synthetic_input(3, 2, (32, 32, 64)).to(device)
    target = (x[:, 0] * 3.999).long().clamp(0, 3)                     # a learnable voxel-wise labelling of the input
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(x), target)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(p.grad is not None for p in net.parameters())
    assert all(b < a for a, b in zip(losses, losses[1:])) and losses[-1] < 0.9 * losses[0], losses


@pytest.mark.parametrize("kw,layers", [
    (dict(dimension=3, input_nc=1, output_nc=32, num_downs=2, ngf=32, norm="instance", pooling="Avg", interp="trilinear",
          norm_eps=1e-2), [3, 13, 20, 27, 34]),                                     # the anatomix-dev recipe, shallow
    (dict(dimension=3, input_nc=1, output_nc=16, num_downs=1, ngf=16, norm="instance_affine", activation="lrelu"),
     [0, 3, 5, 10, 13, 17, 20]),
    (dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=16, norm="batch", pooling="Avg", interp="trilinear"),
     [3, 13, 20, 27, 34]),
])
def test_instance_norm_avgpool_trilinear_variants_train_on_hip(device, kw, layers):
    """The other normalisation / pooling / interpolation modes of the reference's Unet through the differentiable HIP path:
    InstanceNorm3d (per-sample statistics, conv bias), AvgPool3d and trilinear upsampling with their adjoints."""
    hip, ref = _pair_kw(device, kw, "f16")
    x = R.synthetic_input(11, 2, (32, 32, 64)).to(device)
    calls = []
    from anatomix_amd.model import train as TR
    orig = TR.forward_train
    TR.forward_train = lambda *a, **k: calls.append(1) or orig(*a, **k)
    try:
        errs, gerrs, cos = _compare(hip, ref, x, layers, device, loss_scale=4096.0)
    finally:
        TR.forward_train = orig
    assert calls, "the call did not go through anatomix_amd.model.train"
    print("fwd", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) < 1e-2, errs
    # a conv bias in front of an InstanceNorm has a zero gradient (the norm removes the mean): both sides hold rounding noise
    # there, so those entries are checked for smallness instead of direction
    mods = list(hip.model)
    dead = set()
    if kw["norm"].startswith("instance"):
        for i, m in enumerate(mods):
            if isinstance(m, torch.nn.Conv3d) and i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.InstanceNorm3d) \
                    and m.bias is not None and i not in layers:   # (a tap at the conv id sees the pre-norm output: live bias)
                dead.add(f"model.{i}.bias")
    gp = dict(hip.named_parameters())
    for k in dead:
        wk = k.replace(".bias", ".weight")
        assert gp[k].grad.abs().max().item() < 2e-2 * gp[wk].grad.abs().max().item() + 1e-6, k
    live = [k for k in gerrs if k not in dead]
    print("grad worst", max(gerrs[k] for k in live), "min cos", min(cos[k] for k in live))
    last = [k for k in live if k.startswith(f"model.{len(mods) - 1}.")]
    assert all(gerrs[k] < 2e-3 for k in last), {k: gerrs[k] for k in last}
    assert max(gerrs[k] for k in live) < 0.2 and min(cos[k] for k in live) > 0.98


def test_anatomix_dev_variant_takes_a_training_step(device):
    """The 94M InstanceNorm / AvgPool / trilinear variant through forward + backward + AdamW on the HIP path at its smallest
    legal size (64^3: 2^3 at the bottleneck): finite, every parameter receives a gradient, and the loss goes down."""
    kw = R.VARIANTS["anatomix-dev"]
    net = anatomix_amd.Unet(**kw)
    net.load_state_dict(R.synthetic_state_dict(kw, 1, gain=2 ** 0.5))
    net.precision = "bf16"
    net = net.to(device).train()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    x = R.synthetic_input(3, 1, (64, 64, 64)).to(device)
    target = torch.randn(1, 32, 64, 64, 64, device=device) * 0.1
    losses = []
    for _ in range(4):
        opt.zero_grad()
        out = net(x)
        loss = (out - target).square().mean()
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


@pytest.mark.parametrize("mode", ["some_layers_frozen", "whole_network_eval"])
def test_frozen_batchnorm_statistics_under_autograd(device, mode):
    """BatchNorm layers in eval mode inside a differentiated network (the reference freezes the statistics of single layers,
    pretraining/models/base_model.py:175-184; a fully eval-mode network under autograd is the same thing everywhere): the
    running statistics fold into the conv, gradients flow to the conv and to the norm's affine parameters."""
    kw = dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=16, activation="lrelu")
    layers = [3, 13, 20, 27, 34] if mode == "some_layers_frozen" else [4, 13, 21, 34]
    hip, ref = _pair_kw(device, kw, "f16")
    frozen = []
    for net in (hip, ref):
        if mode == "whole_network_eval":
            net.eval()
        else:
            for i, m in enumerate(net.model):
                if isinstance(m, torch.nn.BatchNorm3d) and i % 2 == 0:          # every other norm layer (ids 4, 14, 18, ...)
                    m.eval()
                    if net is hip:
                        frozen.append(i)
    before = {i: hip.model[i].running_mean.clone() for i in frozen}
    x = R.synthetic_input(11, 2, (32, 32, 64)).to(device)
    errs, gerrs, cos = _compare(hip, ref, x, layers, device, loss_scale=4096.0)
    print("fwd", {k: f"{v:.2e}" for k, v in errs.items()}, "grad worst", max(gerrs.values()), "min cos", min(cos.values()))
    assert max(errs.values()) < 1e-2, errs
    last = [k for k in gerrs if k.startswith(f"model.{len(hip.model) - 1}.")]
    assert all(gerrs[k] < 2e-3 for k in last)
    assert max(gerrs.values()) < 0.2 and min(cos.values()) > 0.98, (max(gerrs.values()), min(cos.values()))
    for i in frozen:                                                             # frozen layers keep their statistics
        assert torch.equal(before[i], hip.model[i].running_mean)
        assert int(hip.model[i].num_batches_tracked) == 0


@pytest.mark.parametrize("interp,pooling", [("nearest", "Max"), ("trilinear", "Avg")])
def test_taps_at_pool_and_upsample_ids_are_differentiable(device, interp, pooling):
    """Feature taps at MaxPool / AvgPool ids (the pooled tensor) and at Upsample ids (taken AFTER the concat with the skip,
    network.py:500-502) through the differentiable HIP path."""
    kw = dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=16, interp=interp, pooling=pooling)
    hip, ref = _pair_kw(device, kw, "f16")
    kinds = [type(m).__name__ for m in hip.model]
    layers = [i for i, k in enumerate(kinds) if k in ("MaxPool3d", "AvgPool3d", "Upsample")] + [len(kinds) - 1]
    assert len(layers) == 5
    x = R.synthetic_input(11, 2, (32, 32, 64)).to(device)
    errs, gerrs, cos = _compare(hip, ref, x, layers, device, loss_scale=4096.0)
    print("fwd", {k: f"{v:.2e}" for k, v in errs.items()}, "grad worst", max(gerrs.values()), "min cos", min(cos.values()))
    assert max(errs.values()) < 1e-2, errs
    # The gradient noise floor of comparing two nearby forwards (ReLU / arg-max flips, see the shallow-network test) depends on
    # the configuration -- this Max/nearest one sits at ~0.27 / cos 0.967 with the output tap alone -- so the taps are judged
    # against that floor: they must not add error (a wrong adjoint of a tap is an O(1) change of everything upstream of it).
    hip0, ref0 = _pair_kw(device, kw, "f16")
    _, gerrs0, cos0 = _compare(hip0, ref0, x, [len(kinds) - 1], device, loss_scale=4096.0)
    assert max(gerrs.values()) < max(gerrs0.values()) + 0.05 and min(cos.values()) > min(cos0.values()) - 0.02, \
        (max(gerrs.values()), max(gerrs0.values()), min(cos.values()), min(cos0.values()))
    assert max(gerrs.values()) < 0.35 and min(cos.values()) > 0.95


def test_gradient_with_respect_to_the_input_image(device):
    """A caller that differentiates through the network input (x.requires_grad) gets the stem's data gradient, as the
    reference's stock modules give it -- not None.  (The stem's data-gradient kernel itself is pinned per op in
    tests/test_train_ops_gpu.py, cin = 1 case; here the whole chain runs, so the bound is the one of every gradient that sits
    upstream of ReLU masks: two nearby forwards flip a ~2e-3 fraction of the masks.)"""
    kw = dict(dimension=3, input_nc=1, output_nc=16, num_downs=1, ngf=16)
    hip, ref = _pair_kw(device, kw, "f16")
    x = torch.rand(1, 1, 32, 32, 32, device=device)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    g = torch.randn(1, 16, 32, 32, 32, device=device)
    scale = 64.0                                              # f16 gradients: loss scaling (the cotangent here is O(1) per voxel)
    ((hip(xa) * g).sum() * scale).backward()
    (_ref_forward(ref, xb, []) * g).sum().backward()
    assert xa.grad is not None and xa.grad.shape == x.shape and torch.isfinite(xa.grad).all()
    ga, gb = xa.grad.double().flatten() / scale, xb.grad.double().flatten()
    err = ((ga - gb).norm() / gb.norm()).item()
    cos = (ga @ gb / (ga.norm() * gb.norm())).item()
    print("input gradient: rel err", err, "cos", cos)
    assert err < 0.2 and cos > 0.98, (err, cos)


GOLD128 = np.load(os.path.join(os.path.dirname(__file__), "golden", "pretrain_step128_golden.npz"))


def _step_setup(device, precision, size):
    hip, _ = _pair(device, precision)
    A, B, seg = [t.to(device) for t in PI.step_inputs(size)]
    netF = PatchSampleF(use_mlp=True, nc=PI.NETF_NC, n_mlps=3)
    chans = [128, 256, 128, 64, 32, 16]
    netF.create_mlp([torch.zeros(1, c, 1, 1, 1, device=device) for c in chans])
    netF.load_state_dict(PI.mlp_state_dict(chans, seed=9), strict=True)
    netF = netF.to(device).train()
    opt = Namespace(nce_T=PI.NCE_T, weigh_rarity=False, balance_denominator=False, weighting_mode="raw")
    crits = [SupPatchNCELoss(opt) for _ in PI.NCE_LAYERS]
    return hip, netF, crits, (A, B, seg)


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_contrastive_step_at_128_cube_matches_reference_record(device, precision):
    """BASELINE configs[2] at its REAL size: the record of the reference's own modules (fp32, CPU, oracle/make_golden_pretrain.py
    --step128) -- per-layer losses, total, gradient norms of both networks, and norm + 256 seeded probes of EVERY UNet
    parameter's gradient -- against one step fully on the HIP kernels with the captured coordinates."""
    hip, netF, crits, (A, B, seg) = _step_setup(device, precision, 128)
    ids = [torch.from_numpy(GOLD128[f"step128|ids|{k}"].astype(np.int64)).to(device) for k in range(6)]
    scale = 4096.0 if precision == "f16" else 1.0       # f16 gradients need the usual loss scaling (the reference runs a GradScaler)
    rec = contrastive_step(hip, netF, crits, A, B, seg, PI.NCE_LAYERS, num_patches=PI.NUM_PATCHES, sample_ids=ids, optimizers=None,
                           lambda_nce=scale)
    rec["loss"] /= scale
    rec["grad_norm_G"] /= scale
    rec["grad_norm_F"] /= scale
    for prm in hip.parameters():
        prm.grad /= scale
    got = np.array(list(rec["per_layer"].values()))
    ref = GOLD128["step128|per_layer"]
    dl = np.abs(got - ref) / ref
    dG = abs(rec["grad_norm_G"] - float(GOLD128["step128|grad_norm_G"])) / float(GOLD128["step128|grad_norm_G"])
    dF = abs(rec["grad_norm_F"] - float(GOLD128["step128|grad_norm_F"])) / float(GOLD128["step128|grad_norm_F"])
    print(f"{precision} 128^3: per-layer loss rel dev {dl}, total {abs(rec['loss'] - float(GOLD128['step128|total'])) / rec['loss']:.2e}, "
          f"grad norm G {dG:.2e} F {dF:.2e}")
    worst_n, worst_p = {}, {}
    for name, prm in hip.named_parameters():
        g = prm.grad.double().reshape(-1).cpu()
        rn = float(GOLD128[f"step128|gnorm|{name}"])
        worst_n[name] = abs(g.norm().item() - rn) / rn
        rv = torch.from_numpy(GOLD128[f"step128|gval|{name}"]).double()
        worst_p[name] = rel_l2(g[torch.from_numpy(GOLD128[f"step128|gidx|{name}"])], rv)
    top = sorted(worst_p, key=worst_p.get)[-4:]
    print("  gradient norms: worst", max(worst_n.values()), " probes: worst", {k: f"{worst_p[k]:.2e}" for k in top})
    print("  last layers:", {k: f"{worst_p[k]:.2e}" for k in ("model.65.weight", "model.62.weight", "model.59.weight", "model.0.weight")})
    lim = TRAIN_BOUNDS[precision]
    assert dl.max() <= lim["loss"] and dG <= lim["gnorm"] and dF <= lim["gnorm"], (dl, dG, dF)
    assert max(worst_n.values()) <= lim["layer_gnorm"], worst_n
    assert worst_p["model.65.weight"] <= lim["last_probe"], worst_p["model.65.weight"]


# What one step in 16-bit storage delivers against the reference's fp32 record at 128^3 (measured: bf16 losses 1.4e-4, gradient
# norms 1.2 % / 0.07 %; f16 2.5e-5, 0.9 % / 0.1 %), fixed with ~2x margin.  Single gradient ENTRIES of this random-weight
# network differ far more (the 20 train-mode conv + BatchNorm layers amplify storage rounding ~1.28x per layer, 27 % at the
# output in bf16, and every difference flips ReLU masks): per-parameter norms within 28 % / 20 %, probes of the last layer
# 0.47 / 0.15 relative.  That level is a property of 16-bit storage, not of these kernels -- the reference's own bf16 autocast
# on the stock modules deviates from fp32 just as much, which test_full_6m_network_train_step_sanity asserts side by side.
TRAIN_BOUNDS = {
    "bf16": dict(loss=5e-4, gnorm=2e-2, layer_gnorm=0.45, last_probe=0.7),
    "f16": dict(loss=1e-4, gnorm=2e-2, layer_gnorm=0.3, last_probe=0.25),
}


def test_loss_trajectory_over_adamw_steps_follows_the_stock_modules(device):
    """Six AdamW steps on the same coordinates: the HIP training path (bf16 storage) against the stock torch modules in fp32
    from identical initial weights -- the composition forward / backward / optimizer is pinned, not only its pieces."""
    size, steps = 64, 6
    hip, netF_h, crits, (A, B, seg) = _step_setup(device, "bf16", size)
    ref = copy.deepcopy(hip)
    ref.allow_torch_path, ref._warned = True, True
    ref.forward = lambda x, layers=[], encode_only=False, verbose=False: ref._forward_torch(x, layers, encode_only, verbose)
    netF_r = copy.deepcopy(netF_h)
    for c in crits:
        c.allow_torch_path = True
    ids = [torch.from_numpy(GOLD[f"step|ids|{k}"].astype(np.int64)).to(device) for k in range(6)]
    traj = {}
    for tag, (g, f) in {"hip": (hip, netF_h), "ref": (ref, netF_r)}.items():
        opts = (torch.optim.AdamW(g.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5),
                torch.optim.AdamW(f.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5))
        traj[tag] = [contrastive_step(g, f, crits, A, B, seg, PI.NCE_LAYERS, num_patches=PI.NUM_PATCHES, sample_ids=ids,
                                      optimizers=opts) for _ in range(steps)]
    lh, lr_ = np.array([r["loss"] for r in traj["hip"]]), np.array([r["loss"] for r in traj["ref"]])
    gh, gr = np.array([r["grad_norm_G"] for r in traj["hip"]]), np.array([r["grad_norm_G"] for r in traj["ref"]])
    print("loss hip", lh, "\nloss ref", lr_, "\ngradnorm hip", gh, "\ngradnorm ref", gr)
    assert lr_[-1] < lr_[0] and lh[-1] < lh[0]                       # both train
    np.testing.assert_allclose(lh, lr_, rtol=3e-3)
    np.testing.assert_allclose(lh[0] - lh, lr_[0] - lr_, atol=0.25 * (lr_[0] - lr_[-1]))   # the same descent, step by step
    # Gradient norms: the first step sees identical weights (0.5 % measured); from the third step on the norm of this random-weight
    # network is chaotic -- the fp32 stock modules THEMSELVES differ by up to 15 % between two runs of this test on the same box
    # (2.19 .. 2.60 at the last step over five runs: MIOpen's backward and index_put accumulate in a run-dependent order), and any
    # change of rounding in the bf16 path moves its later norms by as much (2.06 .. 2.25 at the third step over the code paths of
    # round 3).  So: tight where the comparison is deterministic, a band of the chaotic spread afterwards; the losses above hold the
    # trajectory itself to 3e-3.
    np.testing.assert_allclose(gh[:1], gr[:1], rtol=2e-2)
    np.testing.assert_allclose(gh[1:2], gr[1:2], rtol=0.10)
    np.testing.assert_allclose(gh[2:], gr[2:], rtol=0.35)


# One forward + backward of the HIP training path against oracle/train_lowp.py -- the SAME graph with the path's rounding points
# (stored 16-bit activations and gradients, fp32 arithmetic in between) evaluated by torch on the CPU.  Unlike the fp32 records above,
# whose 30-45 % bands are a property of 16-bit storage, this separates KERNEL correctness of the backward from storage precision the way
# forward_lowp does for the forward.  The emulated forward is teacher-forced with the tensors the HIP path stored (this network doubles
# a 1-ulp storage flip per layer: un-forced, the forwards of two correct implementations are 1.4e-2 apart at module 62), so every
# layer's adjoint runs on identical stored operands; what remains is fp32 summation order and the gradient flips it causes.
LOWP_BOUNDS = {"f16": dict(fwd=2e-3, conv_w=2e-3, norm_v=2e-2, whole=2e-3), "bf16": dict(fwd=1.5e-2, conv_w=1.5e-2, norm_v=0.15, whole=1.5e-2)}


@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_backward_matches_the_rounding_point_emulation(device, precision):
    from oracle import train_lowp as TL
    cot_layers = [27, 31, 38, 45, 52]
    size = 64
    hip, _ = _pair(device, precision)
    sd = {k: v.detach().cpu().clone() for k, v in hip.state_dict().items()}
    plan = R.build_plan(**{k: v for k, v in KW.items() if k != "dimension"})
    convs = [i for i, k in enumerate(plan.kinds) if k == "conv"][:-1]
    acts = [i for i, k in enumerate(plan.kinds) if k == "act"]
    layers = sorted(set(convs + acts))                      # every stored tensor: X at the conv ids, Y at the activation ids
    x = torch.from_numpy(np.random.RandomState(3).rand(2, 1, size, size, size).astype(np.float32))
    scale = 1024.0 if precision == "f16" else 1.0           # f16 gradients need loss scaling; the emulation applies the same factor
    out, feats = hip(x.to(device), layers)
    stored = dict(zip(layers, feats))
    g = torch.Generator().manual_seed(5)
    cots = {l: torch.randn(stored[l].shape, generator=g) / stored[l][0].numel() ** 0.5 for l in cot_layers}
    loss = 0.1 * out.square().mean()
    for l in cot_layers:
        loss = loss + (stored[l] * cots[l].to(device)).sum()
    (loss * scale).backward()
    dt = torch.float16 if precision == "f16" else torch.bfloat16
    torch.set_num_threads(32)
    forced = {l: t.detach().cpu() for l, t in stored.items()}
    ref, out_r, taps_r = TL.parameter_gradients(x, sd, KW, cot_layers, [cots[l] * scale for l in cot_layers], out_weight=0.1 * scale,
                                                lowp=dt, forced=forced)
    lim = LOWP_BOUNDS[precision]
    e_out = rel_l2(out.detach().cpu(), out_r)
    print(precision, "output of the forced forward vs HIP %.2e" % e_out)
    assert e_out <= lim["fwd"], e_out
    worst_w, worst_v = ("", 0.0), ("", 0.0)
    ga, gb = [], []
    for k, p in hip.named_parameters():
        a, b = p.grad.detach().double().cpu(), ref[k].double()
        assert torch.isfinite(a).all(), k
        e = float((a - b).norm() / b.norm().clamp_min(1e-30))
        if a.dim() > 1:
            worst_w = max(worst_w, (k, e), key=lambda t: t[1])
        else:
            worst_v = max(worst_v, (k, e), key=lambda t: t[1])
        ga.append(a.flatten()); gb.append(b.flatten())
    ga, gb = torch.cat(ga), torch.cat(gb)
    whole = float((ga - gb).norm() / gb.norm())
    print(precision, f"gradients vs emulation: whole {whole:.2e}, worst conv weight {worst_w}, worst norm vector {worst_v}")
    assert worst_w[1] <= lim["conv_w"], worst_w
    assert worst_v[1] <= lim["norm_v"], worst_v
    assert whole <= lim["whole"], whole


@pytest.mark.parametrize("switch,off,exact", [("RECOMPUTE_ACT", False, True), ("FUSED_FOLD_SPLIT", False, False),
                                             ("SPLIT_CONCAT_DGRAD", 0, False)])
def test_backward_code_paths_agree_parameter_by_parameter(device, switch, off, exact, monkeypatch):
    """Advisor finding (round 3): the round-3 rewrites of the backward (norm adjoint that recomputes the activation's sign from x,
    reflect-padding adjoint fused into the concat split, the 48 -> 16 data gradient as two launches) are switchable, and the only
    end-to-end gradient check past the first step is a 35 % band.  Here each rewrite is held DETERMINISTICALLY: identical weights,
    inputs and cotangents with the switch on and off, every parameter gradient compared on its own -- bit-identical where the
    arithmetic is the same, within bf16 storage rounding of the intermediate gradients where the summation order differs."""
    from anatomix_amd.model import train as TR
    layers = [27, 31, 38, 45, 52, 65]
    x = torch.from_numpy(np.random.RandomState(3).rand(2, 1, 64, 64, 64).astype(np.float32)).to(device)
    grads = {}
    for state in ("on", "off"):
        if state == "off":
            monkeypatch.setattr(TR, switch, off)
        hip, _ = _pair(device, "bf16")
        out, feats = hip(x, layers)
        g = torch.Generator().manual_seed(5)
        loss = 0.1 * out.square().mean()
        for f in feats:
            loss = loss + (f * (torch.randn(f.shape, generator=g).to(device) / f[0].numel() ** 0.5)).sum()
        loss.backward()
        grads[state] = {k: p.grad.detach().clone() for k, p in hip.named_parameters()}
    assert grads["on"].keys() == grads["off"].keys() and len(grads["on"]) > 50
    worst_w, worst_v, worst_cos = 0.0, 0.0, 1.0
    for k in grads["on"]:
        a, b = grads["on"][k].double(), grads["off"][k].double()
        assert torch.isfinite(a).all() and torch.isfinite(b).all(), k
        if exact:
            assert torch.equal(a, b), k
            continue
        err = float((a - b).norm() / b.norm().clamp_min(1e-30))
        cos = float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-30))
        worst_cos = min(worst_cos, cos)
        if a.dim() > 1:                       # conv weights: thousands of entries, each a sum over every voxel
            worst_w = max(worst_w, err)
            assert err <= 2e-2 and cos >= 0.9995, (k, err, cos)
        else:                                 # BatchNorm gain / shift: 16 .. 256 entries, each a CANCELLING sum of the bf16-stored dz over every
            worst_v = max(worst_v, err)       # voxel -- one more rounding of dz on the way (fused fold) or another summation order shows here first
            assert err <= 0.12 and cos >= 0.995, (k, err, cos)
    if not exact:
        ga = torch.cat([grads["on"][k].flatten().double() for k in grads["on"]])
        gb = torch.cat([grads["off"][k].flatten().double() for k in grads["on"]])
        tot = float((ga - gb).norm() / gb.norm())
        print(switch, f"whole gradient rel-L2 {tot:.2e}; worst conv weight {worst_w:.2e}, worst norm vector {worst_v:.2e}, worst cosine {worst_cos:.5f}")
        # measured: 1.2e-2 / 1.1e-2 whole gradient (three bf16 ulps: the gradient passes ~20 bf16-stored tensors), conv weights <= 1.3e-2,
        # norm vectors <= 7.2e-2, cosines >= 0.998
        assert tot <= 2e-2, tot
        assert abs(float(ga.norm() / gb.norm()) - 1.0) <= 5e-3            # the recorded gradient norm: 0.5 %, not a 35 % band


def test_sampled_tap_route_is_bit_identical_to_the_dense_tap_route(device, monkeypatch):
    """contrastive_step hands netF the 512 sampled rows of each tapped tensor (gathered in place, gradients scattered in place:
    model/train.py forward_train_sampled) instead of dense fp32 copies of six feature maps.  Same draws (the generator is consumed in the
    same order), same values (a gather of the same 16-bit storage), same arithmetic in the adjoint (fp32 add, one rounding, at the
    sampled voxels; + 0 elsewhere): losses, sample ids and EVERY parameter gradient must be bit-identical to the dense route."""
    from anatomix_amd.pretraining import step as ST
    from anatomix_amd.model import train as TR
    orig = TR.forward_train_sampled
    res = {}
    for route in ("sampled", "sampled+rows", "dense"):
        monkeypatch.setattr(ST, "_SAMPLED_TAPS", route != "dense")
        # "sampled": the output conv's backward through the dense kernels (rows scattered into a zero gradient volume) -- the
        # bit-identical claim; "sampled+rows": the product default, that backward straight from the rows (amx_conv3d_backward_sampled)
        monkeypatch.setattr(TR, "SPARSE_OUTPUT_TAP", route == "sampled+rows")
        netG, netF, crits, (vA, vB, seg) = _step_setup(device, "bf16", 64)
        calls = []
        monkeypatch.setattr(TR, "forward_train_sampled", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        torch.manual_seed(11)
        r = contrastive_step(netG, netF, crits, vA, vB, seg, PI.NCE_LAYERS, num_patches=512)
        assert bool(calls) == (route != "dense")
        res[route] = (r, {k: p.grad.detach().clone() for k, p in list(netG.named_parameters()) + list(netF.named_parameters())
                          if p.grad is not None})
    (ra, ga), (rc, gc), (rb, gb) = res["sampled"], res["sampled+rows"], res["dense"]
    assert ra["loss"] == rb["loss"] and ra["per_layer"] == rb["per_layer"]
    for a, b in zip(ra["sample_ids"], rb["sample_ids"]):
        assert torch.equal(a, b)
    assert ga.keys() == gb.keys() and len(ga) > 60
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k
    # from the rows: the same forward, the same draws; the output conv's weight gradient is the same sum in another order, its data
    # gradient is rounded once at the voxels next to a face (the dense route rounds the padded domain and the reflect fold separately),
    # which the train-mode BatchNorm layers below spread over every parameter at the level of the storage rounding
    assert rc["loss"] == ra["loss"] and rc["per_layer"] == ra["per_layer"]
    last = max(k for k in ga if k.startswith("model.") and k.endswith(".weight") and ga[k].dim() == 5)
    worst = {"conv": 0.0, "norm": 0.0, "head": 0.0}
    for k in ga:
        e = float((gc[k].double() - ga[k].double()).norm() / ga[k].double().norm().clamp_min(1e-30))
        kind = "head" if k.startswith("mlp_") else ("conv" if ga[k].dim() == 5 else "norm")
        worst[kind] = max(worst[kind], e)
        # (bf16 storage: the bounds of test_backward_matches_the_rounding_point_emulation -- the d gamma / d beta sums cancel)
        assert e < (1e-5 if k == last else {"conv": 1.5e-2, "norm": 0.15, "head": 1e-5}[kind]), (k, e)
    print("sampled+rows vs sampled: worst relative gradient difference", worst)


def test_sampled_route_is_only_taken_where_forward_would_route_to_the_training_function(device):
    """contrastive_step's sampled-tap route calls the training Function directly; it must stand back where ``netG(...)`` would not have
    gone there (eval mode with every parameter frozen: the inference path) and where someone hooked the module's __call__."""
    from anatomix_amd.pretraining import step as ST
    netG, netF, crits, (vA, vB, seg) = _step_setup(device, "bf16", 32)
    reals = torch.cat((vA, vB), 0)
    assert ST._sampled_route(netG, netF, reals, PI.NCE_LAYERS, 512) is not None
    h = netG.register_forward_hook(lambda m, i, o: None)
    assert ST._sampled_route(netG, netF, reals, PI.NCE_LAYERS, 512) is None
    h.remove()
    assert ST._sampled_route(netG, netF, reals, PI.NCE_LAYERS, 512) is not None
    netG.eval()
    for p in netG.parameters():
        p.requires_grad_(False)
    assert ST._sampled_route(netG, netF, reals, PI.NCE_LAYERS, 512) is None
