"""GPU: f16 storage must never hand out silently wrong features.  Folded BatchNorm gains of a real checkpoint are unbounded
and f16 overflows at 65504; behind a ReLU or a max-pool an Inf can even come out finite.  With adversarially scaled
parameters: f16 / f16x2 raise (AmxOverflowError from check_numerics() or from the next call, and the offending forward's
output is all NaN), bf16 and strict (bf16x2) -- fp32's exponent range -- simply give the right answer."""
import numpy as np
import pytest
import torch

from _util import rel_l2
import anatomix_amd
from anatomix_amd._lib import AmxOverflowError
from oracle import unet_ref as R

pytestmark = pytest.mark.gpu
KW = R.VARIANTS["anatomix"]


def _model(device, sd, kw, precision):
    m = anatomix_amd.Unet(**kw)
    m.load_state_dict(sd, strict=True)
    m.precision = precision
    return m.to(device).eval()


def _huge_gain_sd():
    sd = R.synthetic_state_dict(KW, 0)
    sd["model.1.weight"] = sd["model.1.weight"] * 3e5          # BatchNorm gain of the stem: activations ~1e5 from here on
    return sd


@pytest.mark.parametrize("precision", ["f16", "f16x2"])
def test_f16_overflow_is_loud(device, precision):
    sd = _huge_gain_sd()
    m = _model(device, sd, KW, precision)
    x = R.synthetic_input(100, 1, (32, 32, 32)).to(device)
    with torch.no_grad():
        y = m(x)
        torch.cuda.synchronize()
        assert torch.isnan(y).all()                              # never a plausible-looking tensor
        with pytest.raises(AmxOverflowError, match="f16 range"):
            m.check_numerics()
        m.check_numerics()                                       # reported once, then cleared
        y = m(x)                                                 # overflows again ...
        torch.cuda.synchronize()
        with pytest.raises(FloatingPointError):                  # ... and the NEXT call reports it if nobody asked
            m(x)
        # healthy parameters on the same module: clean again
        m.load_state_dict(R.synthetic_state_dict(KW, 0), strict=True)
        y = m(x)
        m.check_numerics()
        assert torch.isfinite(y).all()


@pytest.mark.parametrize("precision,tol", [("bf16", 3e-2), ("strict", 1e-3)])
def test_wide_range_modes_give_the_right_answer(device, precision, tol):
    sd = _huge_gain_sd()
    m = _model(device, sd, KW, precision)
    x = R.synthetic_input(100, 1, (32, 32, 32))
    with torch.no_grad():
        y = m(x.to(device)).cpu()
        m.check_numerics()
        ref = R.forward(x, sd, KW)
    assert torch.isfinite(y).all() and float(ref.abs().max()) > 1e3
    assert rel_l2(y, ref) <= tol, rel_l2(y, ref)


def test_large_negative_preactivations_behind_relu_are_not_an_overflow(device):
    """relu(-1e6) = 0 is the right answer and exactly representable: what is checked is the value about to be stored."""
    sd = R.synthetic_state_dict(KW, 0)
    sd["model.4.bias"] = sd["model.4.bias"] - 1e6               # BatchNorm shift of the second conv: every voxel far below zero
    m = _model(device, sd, KW, "f16")
    x = R.synthetic_input(100, 1, (32, 32, 32))
    with torch.no_grad():
        y = m(x.to(device)).cpu()
        m.check_numerics()
        ref = R.forward(x, sd, KW)
    assert torch.isfinite(y).all() and rel_l2(y, ref) <= 1e-3


def test_instance_norm_variant_and_sliding_window(device):
    """anatomix-dev stores the RAW conv output before the InstanceNorm pass: a huge conv gain overflows there.  The
    sliding-window caller checks once per volume, where the accumulated features leave the library."""
    from anatomix_amd.registration.sliding_window import sliding_window_inference
    kw = dict(dimension=3, input_nc=1, output_nc=16, num_downs=2, ngf=16, norm="instance", pooling="Avg", interp="trilinear",
              norm_eps=1e-2)
    sd = R.synthetic_state_dict(kw, 1)
    sd["model.3.weight"] = sd["model.3.weight"] * 1e6
    x = R.synthetic_input(3, 1, (32, 32, 64))
    with torch.no_grad():
        m = _model(device, sd, kw, "f16")
        m(x.to(device))
        with pytest.raises(AmxOverflowError):
            m.check_numerics()
        with pytest.raises(AmxOverflowError):
            sliding_window_inference(x.to(device), (32, 32, 32), 2, m, overlap=0.5, mode="gaussian")
        ms = _model(device, sd, kw, "strict")
        y = sliding_window_inference(x.to(device), (32, 32, 32), 2, ms, overlap=0.5, mode="gaussian").cpu()
        ref = sliding_window_inference(x, (32, 32, 32), 2, lambda t: R.forward(t, sd, kw), overlap=0.5, mode="gaussian")
    assert rel_l2(y, ref) <= 1e-3, rel_l2(y, ref)
