"""GPU parity of the InstanceNorm / trilinear / AvgPool variant (`anatomix-dev`, BASELINE config 4;
reference anatomix/model/load_from_hf.py:18-24) and of its two bandwidth kernels called through the C ABI.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import TORCH_T, from_ndhwc, max_rel, rel_l2, to_ndhwc
import anatomix_amd
from anatomix_amd import _lib
from oracle import unet_ref as R

pytestmark = pytest.mark.gpu

KW = R.VARIANTS["anatomix-dev"]


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(1, 16, 4, 6, 8), (2, 32, 8, 8, 8), (1, 256, 2, 2, 2), (2, 64, 3, 5, 7)])
def test_upsample2_trilinear_kernel(device, precision, shape):
    lib = _lib.load()
    tdt = TORCH_T[precision]
    n, c, d, h, w = shape
    x = torch.randn(shape, generator=torch.Generator().manual_seed(3))
    dx = to_ndhwc(x, tdt).to(device)
    out = torch.full((n, 2 * d, 2 * h, 2 * w, c), float("nan"), dtype=tdt, device=device)
    _lib.check(lib.amx_upsample2_trilinear(_lib.ptr(dx), _lib.ptr(out), n, d, h, w, c, _lib.PRECISION[precision],
                                           _stream(device)))
    torch.cuda.synchronize(device)
    got = from_ndhwc(out.cpu().float())
    ref = F.interpolate(x.to(tdt).double(), scale_factor=2, mode="trilinear").to(tdt).float()
    # fp32 blend of 8 rounded inputs, one store rounding: at most 1 ulp from the correctly rounded result
    ulp = 2.0 ** (-10 if precision == "f16" else -7)
    assert (got - ref).abs().max().item() <= ulp * ref.abs().max().item()
    assert rel_l2(got, ref) < ulp / 4


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("affine", [False, True])
@pytest.mark.parametrize("n,c,vox,act", [(1, 32, 64 ** 3 // 8, 1), (2, 64, 1000, 2), (1, 1024, 8, 1), (3, 16, 17, 0)])
def test_instance_norm_kernel(device, precision, affine, n, c, vox, act):
    lib = _lib.load()
    tdt = TORCH_T[precision]
    g = torch.Generator().manual_seed(11)
    # offset + scale per channel: exercises the shifted-sum variance (mean >> std for some channels)
    x = torch.randn(n, vox, c, generator=g) * (0.1 + torch.rand(c, generator=g)) + 4.0 * torch.randn(c, generator=g)
    xq = x.to(tdt)
    gamma = (0.5 + torch.rand(c, generator=g)) if affine else None
    beta = torch.randn(c, generator=g) if affine else None
    dx = xq.to(device).contiguous()
    scratch = torch.empty(lib.amx_instance_norm_scratch_bytes(n, c), dtype=torch.uint8, device=device)
    dg = None if gamma is None else gamma.to(device)
    db = None if beta is None else beta.to(device)
    eps = 1e-2
    _lib.check(lib.amx_instance_norm(_lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), eps, n, vox, c, act, 0.3,
                                     _lib.ptr(scratch), _lib.PRECISION[precision], _stream(device)))
    torch.cuda.synchronize(device)
    got = dx.cpu().float()
    xd = xq.double().permute(0, 2, 1)                                    # [n, c, vox]
    ref = F.instance_norm(xd, weight=None if gamma is None else gamma.double(),
                          bias=None if beta is None else beta.double(), eps=eps)
    if act == 1:
        ref = F.relu(ref)
    elif act == 2:
        ref = F.leaky_relu(ref, 0.3)
    ref = ref.permute(0, 2, 1).float()
    ulp = 2.0 ** (-10 if precision == "f16" else -7)
    assert (got - ref).abs().max().item() <= 1.5 * ulp * ref.abs().max().item()
    assert rel_l2(got, ref) < ulp


def _model(device, seed, precision="f16"):
    m = anatomix_amd.Unet(**KW)
    sd = R.synthetic_state_dict(KW, seed)
    m.load_state_dict(sd, strict=True)
    m.precision = precision
    return m.to(device).eval(), sd


@pytest.mark.parametrize("size,n", [((64, 64, 64), 1), ((64, 96, 64), 2)])
def test_dev_forward_matches_oracles(device, size, n):
    m, sd = _model(device, 0)
    x = R.synthetic_input(100, n, size)
    with torch.no_grad():
        assert m.hip_unsupported_reason(x.to(device)) is None
        y = m(x.to(device)).cpu()
        ref_emul = R.forward_lowp(x, sd, KW, torch.float16)
        ref = R.forward(x, sd, KW)
    assert y.shape == ref.shape and torch.isfinite(y).all()
    e_emul, e_ref = rel_l2(y, ref_emul), rel_l2(y, ref)
    print(f"anatomix-dev {size} n={n}: vs emulated {e_emul:.3e}  vs fp32 {e_ref:.3e}  max-rel {max_rel(y, ref):.3e}")
    # InstanceNorm over the 2^3..4^3-voxel planes of the deep levels amplifies ANY 10-bit-mantissa operand
    # rounding on these synthetic weights: the CPU emulation of f16 storage sits at 1.1e-2 rel-L2 from fp32, and
    # rounding ONLY the conv operands (what cuDNN's TF32 path of the reference does) already gives 8.5e-3
    # (DESIGN.md "anatomix-dev numerics").  The kernel-correctness distance is the one to the emulation.
    assert e_emul < 1.2e-2, e_emul
    assert e_ref < 2e-2, e_ref


def test_dev_forward_golden_probes(device):
    """Golden vectors generated from the real reference (oracle/make_golden.py): anatomix-dev, 64^3, seed 0."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_forward_golden.npz"))
    for seed in (0, 1):
        tag = f"anatomix-dev|s{seed}|64|g1.0000"
        m, sd = _model(device, seed)
        x = R.synthetic_input(100 + seed, 1, (64, 64, 64))
        with torch.no_grad():
            y = m(x.to(device)).cpu()
        vals = torch.from_numpy(g[tag + "|val"]).float()
        got = y.flatten()[torch.from_numpy(g[tag + "|idx"]).long()]
        assert rel_l2(got, vals) < 2e-2, rel_l2(got, vals)
        v = y.transpose(0, 1).reshape(y.shape[1], -1).double()
        np.testing.assert_allclose(v.norm(dim=1).numpy(), g[tag + "|stats"][2], rtol=2e-2)


def test_dev_batch_independence_and_determinism(device):
    m, _ = _model(device, 1)
    x = R.synthetic_input(5, 2, (64, 64, 64)).to(device)
    with torch.no_grad():
        y2 = m(x)
        y0 = m(x[:1])
        y2b = m(x)
    assert torch.equal(y2[:1], y0)
    assert torch.equal(y2, y2b)


@pytest.mark.parametrize("kw", [
    dict(dimension=3, input_nc=1, output_nc=16, num_downs=3, ngf=16, norm="instance_affine", interp="trilinear",
         pooling="Avg"),
    dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=16, norm="instance", interp="nearest",
         pooling="Max", norm_eps=1e-2, activation="lrelu"),
    dict(dimension=3, input_nc=1, output_nc=32, num_downs=2, ngf=32, norm="batch", interp="trilinear",
         pooling="Avg", use_skip_connection=False),
])
def test_mixed_mode_networks(device, kw):
    """Every norm / interp / pooling combination goes through the same schedule: spot-check three mixes."""
    m = anatomix_amd.Unet(**kw)
    sd = R.synthetic_state_dict(kw, 3)
    m.load_state_dict(sd, strict=True)
    m = m.to(device).eval()
    x = R.synthetic_input(7, 2, (32, 48, 32))
    with torch.no_grad():
        y = m(x.to(device)).cpu()
        ref_emul = R.forward_lowp(x, sd, kw, torch.float16)
        ref = R.forward(x, sd, kw)
    print(kw["norm"], kw["interp"], rel_l2(y, ref_emul), rel_l2(y, ref))
    # instance-norm nets amplify 1-ulp storage flips (see test_dev_forward_matches_oracles): the HIP result must be
    # as close to fp32 as the emulation is, and within a few 1e-3 of the emulation itself
    e_emul_ref = rel_l2(ref_emul, ref)
    assert rel_l2(y, ref_emul) < 6e-3, rel_l2(y, ref_emul)
    assert rel_l2(y, ref) < max(1.5 * e_emul_ref, 1e-3), (rel_l2(y, ref), e_emul_ref)
