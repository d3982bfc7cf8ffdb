"""GPU parity of Unet.forward(input, layers, encode_only) (reference network.py:475-529) on the HIP path."""
import os

import numpy as np
import pytest
import torch

from _util import max_rel, rel_l2
import anatomix_amd
from oracle import unet_ref as R

pytestmark = pytest.mark.gpu

TAPG = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_taps_golden.npz"))
TOL = {"anatomix": 1.5e-3, "anatomix-dev": 2e-2}      # f16 storage; see test_unet_dev_gpu.py for the dev figure


def _model(device, variant, seed=0):
    kw = R.VARIANTS[variant]
    m = anatomix_amd.Unet(**kw)
    sd = R.synthetic_state_dict(kw, seed)
    m.load_state_dict(sd, strict=True)
    return m.to(device).eval(), sd, kw


@pytest.mark.parametrize("variant,size", [("anatomix", 32), ("anatomix-dev", 64)])
def test_taps_match_reference_golden(device, variant, size):
    tag = f"{variant}|s0|{size}"
    taps = [int(t) for t in TAPG[tag + "|taps"]]
    m, sd, kw = _model(device, variant)
    x = R.synthetic_input(100, 1, (size,) * 3)
    with torch.no_grad():
        y, feats = m(x.to(device), taps)
        y_plain = m(x.to(device))
    assert len(feats) == len(taps)
    for t, f in zip(taps, feats):
        f = f.cpu()
        assert list(f.shape) == list(TAPG[tag + f"|tap{t}|shape"]), t
        got = f.reshape(-1)[torch.from_numpy(TAPG[tag + f"|tap{t}|idx"])]
        ref = torch.from_numpy(TAPG[tag + f"|tap{t}|val"])
        assert rel_l2(got, ref) < TOL[variant], (t, rel_l2(got, ref))
    # tapping pre-norm outputs re-routes BatchNorm through its own pass: same result up to storage rounding
    assert rel_l2(y.cpu(), y_plain.cpu()) < TOL[variant]


def test_pretraining_tap_set_128(device):
    """The nce_layers of the reference's pretraining launcher (pretrain_anatomix.py:385) at 2 x 128^3."""
    layers = [27, 31, 38, 45, 52, 65]
    m, sd, kw = _model(device, "anatomix")
    x = R.synthetic_input(100, 2, (128, 128, 128))
    with torch.no_grad():
        y, feats = m(x.to(device), layers, False)
        ry, rf = R.forward(x, sd, kw, layers=layers)
    shapes = [(2, 128, 16, 16, 16), (2, 256, 8, 8, 8), (2, 128, 16, 16, 16), (2, 64, 32, 32, 32), (2, 32, 64, 64, 64),
              (2, 16, 128, 128, 128)]
    assert [tuple(f.shape) for f in feats] == shapes
    for l, f, r in zip(layers, feats, rf):
        assert rel_l2(f.cpu(), r) < 1.5e-3, (l, rel_l2(f.cpu(), r))
    assert rel_l2(y.cpu(), ry) < 1e-3
    assert torch.equal(feats[-1], y)            # the tap at the output conv IS the output


@pytest.mark.parametrize("layers,encode_only", [
    ([31, 27], False),          # collected in traversal order, whatever the order of `layers`
    ([31, 27], True),           # ... and encode_only stops at layers[-1] = 27: the tap at 31 is never reached
    ([0], True), ([1], True), ([2], True),    # stop inside the first conv/norm/act group: later modules must not run
    ([9, 37, 58], False), ([3, 4, 5, 8, 9], True), ([64, 65], False), ([44], True),
])
def test_layers_and_encode_only_semantics(device, layers, encode_only):
    m, sd, kw = _model(device, "anatomix", seed=1)
    x = R.synthetic_input(7, 1, (32, 32, 32))
    with torch.no_grad():
        got = m(x.to(device), layers, encode_only)
        ref = R.forward(x, sd, kw, layers=layers, encode_only=encode_only)
    if encode_only:
        gf, rf = got, ref
    else:
        (gy, gf), (ry, rf) = got, ref
        assert rel_l2(gy.cpu(), ry) < 1e-3
    assert len(gf) == len(rf)
    for a, b in zip(gf, rf):
        assert a.shape == b.shape
        assert rel_l2(a.cpu(), b) < 1.5e-3, (layers, rel_l2(a.cpu(), b), max_rel(a.cpu(), b))


def test_taps_for_no_norm_lrelu_and_instance_nets(device):
    for kw, layers in [
        (dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=16, norm="none", activation="lrelu"), [0, 1, 5, 13, 14]),
        (dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=16, norm="instance_affine", interp="trilinear",
              pooling="Avg"), [0, 1, 2, 9, 23, 24, 25]),
    ]:
        m = anatomix_amd.Unet(**kw)
        sd = R.synthetic_state_dict(kw, 2)
        m.load_state_dict(sd, strict=True)
        m = m.to(device).eval()
        x = R.synthetic_input(3, 2, (32, 32, 32))
        with torch.no_grad():
            gy, gf = m(x.to(device), layers)
            ry, rf = R.forward(x, sd, kw, layers=layers)
            ge = m(x.to(device), layers[:2], True)
            re_ = R.forward(x, sd, kw, layers=layers[:2], encode_only=True)
        for l, a, b in zip(layers, gf, rf):
            assert a.shape == b.shape and rel_l2(a.cpu(), b) < 6e-3, (kw["norm"], l, rel_l2(a.cpu(), b))
        for a, b in zip(ge, re_):
            assert rel_l2(a.cpu(), b) < 6e-3
