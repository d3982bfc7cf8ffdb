"""GPU: the stem-fed z-march launch (network.py modules 0..5 of the 6 M model as ONE kernel: the stem's 16-channel tensor never
reaches HBM; amx_conv3d_zmarch.hip, STEM) against the two-launch route.  The fused kernel forms the stem's values with the same
operands, the same MFMA sequence and the same rounding, so the two routes must agree BIT FOR BIT -- on every face of the volume
(reflected stem OUTPUT in the ring halo), for tiles in every position, batches, both 16-bit storage types, feature taps behind the
pair, and sliding-window batches (each window is reflect-padded on its own).  A feature tap inside the pair (module 2) is what
selects the two-launch route in the product library; the whole-forward parity against the CPU oracle is tests/test_unet_gpu.py,
which runs the fused launch."""
import pytest
import torch

import anatomix_amd
from anatomix_amd._lib import AmxOverflowError
from oracle import unet_ref as R

pytestmark = pytest.mark.gpu
KW = R.VARIANTS["anatomix"]


def _model(device, precision, seed=0):
    m = anatomix_amd.Unet(**KW)
    m.load_state_dict(R.synthetic_state_dict(KW, seed), strict=True)
    m.precision = precision
    return m.to(device).eval()


def _kernels(m, x):
    _, recs = m.profile_forward(x)
    return [r["kernel"] for r in recs]


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("n,size", [(1, (32, 32, 32)), (2, (64, 64, 64)), (1, (48, 64, 96)), (3, (80, 48, 32)), (1, (32, 128, 64))])
def test_fused_equals_two_launches_bit_for_bit(device, precision, n, size):
    m = _model(device, precision)
    x = R.synthetic_input(100 + n, n, size).to(device)
    with torch.no_grad():
        assert "stem1->16->16" in _kernels(m, x)[0]             # the plain forward takes the fused launch ...
        y = m(x)
        y2, feats = m.forward_hip_taps(x, [2])                   # ... a tap on the stem's tensor the two launches
    assert torch.isfinite(y).all()
    assert torch.equal(y, y2)
    assert feats[0].shape == (n, 16) + size


def test_fused_headline_shape_bit_for_bit(device):
    m = _model(device, "f16", seed=1)
    x = R.synthetic_input(7, 4, (128, 128, 128)).to(device)
    with torch.no_grad():
        y = m(x)
        y2, _ = m.forward_hip_taps(x, [2])
    assert torch.equal(y, y2)


def test_taps_behind_the_pair_keep_the_fused_launch(device):
    m = _model(device, "f16")
    x = R.synthetic_input(5, 1, (32, 64, 64)).to(device)
    with torch.no_grad():
        ya, fa = m.forward_hip_taps(x, [5, 8])                   # module 5: the pair's activated output (fused launch)
        yb, fb = m.forward_hip_taps(x, [2, 5, 8])                # module 2 forces the two launches
        enc = m.forward_hip_taps(x, [5], encode_only=True)       # stops behind the pair
    assert torch.equal(ya, yb)
    assert torch.equal(fa[0], fb[1]) and torch.equal(fa[1], fb[2])
    assert len(enc) == 1 and torch.equal(enc[0], fa[0])


def test_window_batches_are_padded_per_window(device):
    """amx_unet_forward_windows (batches of windows at arbitrary x / y starts through the fused launch): every window is its own
    reflect-padded input of the stem.  Composition of one-window plain forwards with the same importance map and order."""
    from anatomix_amd.registration import sliding_window as SW
    m = _model(device, "f16")
    vol = R.synthetic_input(11, 1, (64, 96, 89)).to(device)
    roi = (64, 64, 64)
    with torch.no_grad():
        got = SW.sliding_window_inference(vol, roi, 4, m, overlap=0.6, mode="gaussian")
        starts = SW.window_starts(vol.shape[2:], roi, 0.6)
        wmap = SW.importance_map(roi, "gaussian", 0.125, device)
        acc = torch.zeros_like(got)
        cnt = torch.zeros((1, 1) + tuple(vol.shape[2:]), device=device)
        for (z, y, x) in starts:
            win = vol[:, :, z:z + 64, y:y + 64, x:x + 64].contiguous()
            acc[:, :, z:z + 64, y:y + 64, x:x + 64] += wmap * m(win)
            cnt[:, :, z:z + 64, y:y + 64, x:x + 64] += wmap
        want = acc / cnt
    assert len(starts) > 4 and any(s[2] % 4 for s in starts)      # several batches, window starts that are not 16-byte aligned
    err = ((got - want).abs().max() / want.abs().max()).item()
    assert err < 1e-5, err                                        # (fp32 accumulation order of the weighted sums only)


@pytest.mark.parametrize("pattern", [0x7E007E00, 0x7C00FC00, 0xFFFFFFFF])
def test_results_do_not_depend_on_what_the_previous_kernel_left_in_lds(device, pattern):
    """The pad elements of the input row copies are read against zero weights: with NaN / Inf bit patterns left in a CU's LDS by the
    kernel before, 0 x Inf = NaN reached a few voxels of the fused launch's output on some boxes (once, in the registration test) until
    the input ring was zero-filled at kernel start.  amx_debug_fill_lds writes the pattern into every CU's LDS right before the forward."""
    import ctypes
    from anatomix_amd import _lib
    lib = _lib.load()
    m = _model(device, "f16")
    x = R.synthetic_input(21, 2, (64, 64, 64)).to(device)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    with torch.no_grad():
        want, _ = m.forward_hip_taps(x, [2])
        for _ in range(3):
            _lib.check(lib.amx_debug_fill_lds(pattern, st))
            y = m(x)
            assert torch.equal(y, want)
            _lib.check(lib.amx_debug_fill_lds(pattern, st))
            y2, _ = m.forward_hip_taps(x, [2])                  # (the two-launch route and every other kernel of the forward, too)
            assert torch.equal(y2, want)
    m.check_numerics()


def test_input_beyond_f16_is_loud(device):
    """The fused path rounds the network input to the storage type in its preparation pass: a value beyond the f16 range must raise
    like an overflowing activation does (tests/test_range_safety_gpu.py covers the stem's own outputs through the fused launch)."""
    m = _model(device, "f16")
    x = R.synthetic_input(3, 1, (32, 32, 32)).to(device)
    x[0, 0, 5, 6, 7] = 1.0e5
    with torch.no_grad():
        y = m(x)
        torch.cuda.synchronize()
        assert torch.isnan(y).all()
        with pytest.raises(AmxOverflowError, match="f16 range"):
            m.check_numerics()
        x[0, 0, 5, 6, 7] = 1.0
        y = m(x)
        m.check_numerics()
        assert torch.isfinite(y).all()
