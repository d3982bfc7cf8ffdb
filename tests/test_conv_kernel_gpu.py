"""GPU parity of the single-layer entry amx_conv3d_k3_reflect against a CPU fp64-accumulated
reference fed the same fp16/bf16-rounded operands.  Covers every tile configuration the
launcher can pick (W >= 32 / 16 / <= 8 x Q in {1,2,4}), reflect borders, partial bricks, the
fused nearest-upsample + concat input and the fp32 planar epilogue."""
import numpy as np
import pytest
import torch

from _util import TORCH_T, max_rel, ref_conv, ref_conv_fp64, ref_conv_upcat_merged, rel_l2, run_conv, run_conv_merged

pytestmark = pytest.mark.gpu

CASES = [
    # (c0, c1, cout, (d,h,w), n, act)   -- shapes from SURVEY.md section 8c(iii) plus edge cases
    (16, 0, 16, (32, 32, 32), 1, 1),
    (16, 0, 16, (8, 16, 32), 2, 0),
    (16, 0, 32, (16, 16, 32), 1, 1),      # Q=2, W>=32
    (32, 0, 32, (8, 8, 64), 1, 1),
    (16, 32, 16, (16, 16, 32), 1, 1),     # skip || up  (48 -> 16): merged-tap z-march kernel
    (16, 32, 16, (12, 24, 64), 2, 1),     # same, ragged tile rows, batch 2
    (16, 32, 16, (6, 6, 16), 1, 1),       # same shape class below the merged kernel's minimum: generic path
    (32, 64, 32, (8, 8, 32), 1, 1),       # 96 -> 32
    (64, 0, 64, (8, 8, 32), 1, 1),        # Q=4, W>=32
    (32, 0, 64, (8, 8, 16), 1, 1),        # W=16 class
    (64, 128, 64, (8, 8, 16), 1, 2),      # 192 -> 64 @16, leaky relu
    (128, 0, 128, (16, 16, 16), 1, 1),    # Q=2 NCH=2 @16
    (128, 256, 128, (4, 8, 16), 1, 1),    # 384 -> 128
    (128, 0, 256, (8, 8, 8), 1, 1),       # bottleneck 8^3, Q=1 NCH=4
    (256, 0, 256, (8, 8, 8), 2, 1),
    (48, 0, 16, (6, 10, 20), 1, 1),       # ragged: partial bricks on every axis, 3 chunks
    (16, 0, 16, (2, 2, 2), 1, 1),         # minimum legal size for reflect
    (64, 0, 128, (4, 4, 4), 1, 1),        # tiny level of a 32^3 network
    (16, 0, 16, (33, 17, 35), 1, 1),      # odd sizes
]


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "c%d+%d_o%d_%dx%dx%d_n%d_a%d" % (c[0], c[1], c[2], *c[3], c[4], c[5]))
def test_conv_matches_cpu(device, case, precision):
    c0, c1, cout, (d, h, w), n, act = case
    rs = np.random.RandomState(hash((c0, c1, cout, d, h, w)) & 0xFFFF)
    x0 = torch.from_numpy(rs.randn(n, c0, d, h, w).astype(np.float32))
    x1 = None
    if c1:
        x1 = torch.from_numpy(rs.randn(n, c1, d // 2, h // 2, w // 2).astype(np.float32))
    wgt = torch.from_numpy((rs.randn(cout, c0 + c1, 3, 3, 3) / np.sqrt(27.0 * (c0 + c1))).astype(np.float32))
    scale = torch.from_numpy(rs.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = torch.from_numpy((rs.randn(cout) * 0.1).astype(np.float32))
    got = run_conv(device, x0, x1, wgt, scale, shift, act, precision)
    merged = (c0, c1, cout) == (16, 32, 16) and w >= 32 and h >= 8 and d >= 4
    ref = (ref_conv_upcat_merged if merged else ref_conv)(x0, x1, wgt, scale, shift, act, precision)
    assert torch.isfinite(got).all()
    # output is stored in the 16-bit type: allow one rounding of the result
    ulp = 2.0 ** -10 if precision == "f16" else 2.0 ** -7
    err = (got.double() - ref.double()).abs()
    tol = ulp * ref.abs().double() + 1e-3 * ulp + 2e-5
    assert (err <= tol).all(), f"max err {err.max().item():.3e} rel_l2 {rel_l2(got, ref):.3e}"


@pytest.mark.parametrize("precision", ["f16x2", "bf16x2"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "c%d+%d_o%d_%dx%dx%d_n%d_a%d" % (c[0], c[1], c[2], *c[3], c[4], c[5]))
def test_conv_strict_precision_matches_fp64(device, case, precision):
    """Strict precision: operands stored as hi + lo pairs ([hi(C) | lo(C)] per voxel), three MFMAs per product, the fp32
    result split into hi + lo again.  Reference: fp64 convolution of the SAME (hi + lo) operand values.  What remains is the
    dropped Wl * xl term (2^-16 / 2^-22 relative), fp32 accumulation, and the split rounding of the output."""
    c0, c1, cout, (d, h, w), n, act = case
    rs = np.random.RandomState(hash((c0, c1, cout, d, h, w)) & 0xFFFF)
    x0 = torch.from_numpy(rs.randn(n, c0, d, h, w).astype(np.float32))
    x1 = torch.from_numpy(rs.randn(n, c1, d // 2, h // 2, w // 2).astype(np.float32)) if c1 else None
    wgt = torch.from_numpy((rs.randn(cout, c0 + c1, 3, 3, 3) / np.sqrt(27.0 * (c0 + c1))).astype(np.float32))
    scale = torch.from_numpy(rs.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = torch.from_numpy((rs.randn(cout) * 0.1).astype(np.float32))
    got = run_conv(device, x0, x1, wgt, scale, shift, act, precision)
    ref = ref_conv(x0, x1, wgt, scale, shift, act, precision)
    assert torch.isfinite(got).all()
    tol = 2e-5 if precision == "bf16x2" else 4e-6      # f16x2: what is left is fp32 accumulation over up to 10368 terms
    assert rel_l2(got, ref) < tol and max_rel(got, ref) < 4 * tol, (rel_l2(got, ref), max_rel(got, ref))
    # and against the un-rounded fp32 operands: the whole point of the mode
    full = ref_conv_fp64(x0, x1, wgt, scale, shift, act)
    assert rel_l2(got, full) < (3e-5 if precision == "bf16x2" else 4e-6), rel_l2(got, full)


@pytest.mark.parametrize("precision", ["f16", "bf16", "f16x2", "bf16x2"])
@pytest.mark.parametrize("cout,c0,c1", [(16, 16, 0), (32, 32, 0), (16, 16, 32)])
def test_conv_planar_fp32_output(device, cout, c0, c1, precision):
    """Final-layer epilogue: fp32 NCDHW, no output rounding -> tight tolerance."""
    rs = np.random.RandomState(7)
    d, h, w = 8, 12, 40
    x0 = torch.from_numpy(rs.randn(2, c0, d, h, w).astype(np.float32))
    x1 = torch.from_numpy(rs.randn(2, c1, d // 2, h // 2, w // 2).astype(np.float32)) if c1 else None
    wgt = torch.from_numpy((rs.randn(cout, c0 + c1, 3, 3, 3) / np.sqrt(27.0 * (c0 + c1))).astype(np.float32))
    got = run_conv(device, x0, x1, wgt, None, None, 0, precision, planar=True)
    ref = (ref_conv_upcat_merged if (c1 and precision in ("f16", "bf16")) else ref_conv)(x0, x1, wgt, None, None, 0, precision)
    assert torch.isfinite(got).all()
    assert max_rel(got, ref) < (4e-5 if precision == "bf16x2" else 2e-5), max_rel(got, ref)


def test_conv_transpose_detecting(device):
    """A = I style check with an asymmetric operand: one-hot weight moves channel 3 at tap
    (kz,ky,kx) = (0,1,2) to output channel 5 -- catches swapped MFMA rows/cols or tap order."""
    c, d, h, w = 16, 4, 6, 32
    rs = np.random.RandomState(3)
    x0 = torch.from_numpy(rs.randn(1, c, d, h, w).astype(np.float32))
    wgt = torch.zeros(16, c, 3, 3, 3)
    wgt[5, 3, 0, 1, 2] = 1.0
    got = run_conv(device, x0, None, wgt, None, None, 0, "f16")
    ref = ref_conv(x0, None, wgt, None, None, 0, "f16")
    assert torch.equal(got, ref.half().float())


MERGED_CASES = [
    # (c0 = cout, c1, (d,h,w), n, act): the wider nearest-upsample concat layers as amx_unet_forward runs them
    (32, 64, (8, 8, 32), 1, 1),           # 96 -> 32, one brick row
    (32, 64, (16, 24, 64), 2, 1),         # z-march skip conv, several bricks, batch 2
    (32, 64, (6, 10, 40), 1, 2),          # ragged: low-res 3 x 5 x 20 (partial tiles on every axis), leaky relu
    (64, 128, (8, 16, 32), 1, 1),         # 192 -> 64: generic skip conv (Q = 4), two cout groups in the merged launch
    (64, 128, (4, 12, 48), 1, 0),         # ragged, no activation
    (128, 256, (4, 8, 32), 1, 1),         # 384 -> 128
    (48, 96, (4, 8, 32), 1, 1),           # cout not a multiple of 32: 16-channel groups in the merged launch
    (128, 256, (8, 8, 16), 2, 1),         # 384 -> 128 @16: low-res rows of 8 cells -> tiles of 8 cells x 2 rows
    (64, 128, (4, 12, 24), 1, 2),         # low-res rows of 12 cells: two 8-cell tiles, the second half empty; odd row count (6)
    (32, 32, (6, 6, 20), 1, 1),           # a single 32-channel stage, low-res 3 x 3 x 10
]


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("case", MERGED_CASES, ids=lambda c: "c%d+%d_%dx%dx%d_n%d_a%d" % (c[0], c[1], *c[2], c[3], c[4]))
def test_merged_concat_conv_matches_cpu(device, case, precision):
    """amx_conv3d_upcat_merged (= how amx_unet_forward runs network.py:403-435 for the wider decoder blocks): merged 2x2x2 taps
    over the low-res tensor, partial sums rounded once to the storage type, skip conv adds them.  Reference: an independent CPU
    formulation of the same algebra in fp64 on the same rounded operands; and the plain 27-tap reference within what the one
    extra rounding of the partial sums can cost."""
    c0, c1, (d, h, w), n, act = case
    cout = c0
    rs = np.random.RandomState(hash((c0, c1, d, h, w)) & 0xFFFF)
    x0 = torch.from_numpy(rs.randn(n, c0, d, h, w).astype(np.float32))
    x1 = torch.from_numpy(rs.randn(n, c1, d // 2, h // 2, w // 2).astype(np.float32))
    wgt = torch.from_numpy((rs.randn(cout, c0 + c1, 3, 3, 3) / np.sqrt(27.0 * (c0 + c1))).astype(np.float32))
    scale = torch.from_numpy(rs.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = torch.from_numpy((rs.randn(cout) * 0.1).astype(np.float32))
    got = run_conv_merged(device, x0, x1, wgt, scale, shift, act, precision)
    ref = ref_conv_upcat_merged(x0, x1, wgt, scale, shift, act, precision, round_partial=True)
    assert torch.isfinite(got).all()
    ulp = 2.0 ** -10 if precision == "f16" else 2.0 ** -7
    err = (got.double() - ref.double()).abs()
    # one rounding of the result + (rarely) a partial sum that rounds the other way after fp32 accumulation-order differences
    # (a flipped partial sum costs one ulp of a value of magnitude <= ~4)
    tol = ulp * (ref.abs().double() + 4.0) + 2e-5
    assert (err <= tol).all(), f"max err {err.max().item():.3e} rel_l2 {rel_l2(got, ref):.3e}"
    flips = (err > ulp * ref.abs().double() + 1e-3 * ulp + 2e-5).double().mean().item()
    assert flips < 2e-3, flips                                             # ... and is rare
    assert rel_l2(got, ref) < (4e-4 if precision == "f16" else 3.2e-3)     # the result's own rounding: 2^-11 / sqrt(3), 2^-8 / sqrt(3)
    plain = ref_conv(x0, x1, wgt, scale, shift, act, precision)          # 27 separately rounded taps, no partial rounding
    assert rel_l2(got, plain) < (6e-4 if precision == "f16" else 5e-3), rel_l2(got, plain)


@pytest.mark.parametrize("precision", ["f16x2", "bf16x2"])
@pytest.mark.parametrize("case", MERGED_CASES + [(16, 32, (8, 16, 32), 1, 1), (16, 32, (12, 24, 64), 2, 1), (16, 32, (4, 4, 16), 1, 1)],
                         ids=lambda c: "c%d+%d_%dx%dx%d_n%d_a%d" % (c[0], c[1], *c[2], c[3], c[4]))
def test_merged_concat_conv_strict_precision_matches_fp64(device, case, precision):
    """The same two-launch form in the strict precisions (hi + lo operands, three MFMAs per product, split partial sums and
    outputs), incl. the 48 -> 16 layer, which has no fused kernel there.  Reference: fp64 convolution of the un-rounded fp32
    operands -- the whole point of the mode; the merged weights are sums formed in fp32 and split once."""
    c0, c1, (d, h, w), n, act = case
    cout = c0
    rs = np.random.RandomState(hash((c0, c1, d, h, w)) & 0xFFFF)
    x0 = torch.from_numpy(rs.randn(n, c0, d, h, w).astype(np.float32))
    x1 = torch.from_numpy(rs.randn(n, c1, d // 2, h // 2, w // 2).astype(np.float32))
    wgt = torch.from_numpy((rs.randn(cout, c0 + c1, 3, 3, 3) / np.sqrt(27.0 * (c0 + c1))).astype(np.float32))
    scale = torch.from_numpy(rs.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = torch.from_numpy((rs.randn(cout) * 0.1).astype(np.float32))
    got = run_conv_merged(device, x0, x1, wgt, scale, shift, act, precision)
    assert torch.isfinite(got).all()
    full = ref_conv_fp64(x0, x1, wgt, scale, shift, act)
    tol = 3e-5 if precision == "bf16x2" else 4e-6
    assert rel_l2(got, full) < tol and max_rel(got, full) < 4 * tol, (rel_l2(got, full), max_rel(got, full))


@pytest.mark.parametrize("shape,limit_us", [((32, 0, 32, 64, 4), 350.0), ((16, 0, 16, 128, 2), 400.0), ((16, 32, 16, 64, 4), 300.0),
                                            ((64, 0, 64, 32, 4), 250.0)])
def test_hot_layers_are_not_pathologically_slow(device, shape, limit_us):
    """A guard, not a benchmark: each of these layers takes 25-80 us on an MI355X, and the limit is ~5x that.  It exists because the
    z-march kernels keep 14-28 weight fragments in registers and a change that makes hipcc index them dynamically (a loop left partly
    rolled) or compile for the wrong occupancy moves them to scratch memory: the results stay bit-identical and the layer takes 10x
    as long (seen twice in round 5: 70 -> 890 us)."""
    import ctypes
    from anatomix_amd import _lib
    c0, c1, cout, S, n = shape
    lib = _lib.load()
    x0 = torch.randn(n, S, S, S, c0, device=device).half()
    x1 = torch.randn(n, S // 2, S // 2, S // 2, c1, device=device).half() if c1 else None
    w = (torch.randn(cout, c0 + c1, 27, device=device) / (27 * (c0 + c1)) ** 0.5).float()
    sh = torch.zeros(cout, device=device)
    wpk = torch.empty(lib.amx_conv3d_packed_bytes(c0 + c1, cout), dtype=torch.uint8, device=device)
    out = torch.empty(n, S, S, S, cout, device=device, dtype=torch.half)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    sb = lib.amx_conv3d_scratch_bytes(c0, c1, cout, n, S, S, S, 0)
    scratch = torch.empty(max(sb, 4) // 4, device=device) if sb else None

    def run():
        _lib.check(lib.amx_conv3d_k3_reflect_ws(_lib.ptr(x0), c0, _lib.ptr(x1), c1, _lib.ptr(w), None, _lib.ptr(sh), cout, n, S, S, S,
                                                1, 0.3, 0, _lib.ptr(wpk), _lib.ptr(out), None, _lib.ptr(scratch), sb, st))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    assert us < limit_us, f"{shape}: {us:.0f} us per launch (limit {limit_us:.0f}): registers spilled to scratch memory?"
