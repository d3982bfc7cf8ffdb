"""GPU parity of the deep-level kernel conv3d_k3_ks (amx_conv3d_ks.hip: weights stationary in registers, K split over the waves,
fixed-order reduction; optional cross-workgroup split-K through caller scratch) against the CPU fp64-accumulated reference on the
same 16-bit-rounded operands -- every register configuration (Q x chunks per wave x K waves x teams), both brick shapes
(W >= 16 / W = 8), partial bricks, brick runs with an idle team, the split-K path with and without scratch, and determinism."""
import numpy as np
import pytest
import torch

from _util import ref_conv, rel_l2, run_conv

pytestmark = pytest.mark.gpu

CASES = [
    # (c0, cout, (d, h, w), n, act)                        configuration the launcher picks (batch-4 sizes of the 6 M network marked *)
    (64, 64, (8, 8, 32), 1, 1),        # Q4 k4x1, brick 2x4x16
    (64, 64, (32, 32, 32), 4, 1),      # * m20 / m48: four bricks per workgroup
    (64, 64, (6, 10, 40), 2, 2),       # Q4 k4x1, partial bricks on every axis, leaky relu
    (32, 64, (8, 8, 32), 1, 1),        # Q4 k2x1, two teams
    (32, 64, (2, 4, 48), 1, 1),        # Q4 k2x1, three bricks: a run with an idle second team
    (32, 64, (32, 32, 32), 4, 1),      # * m17
    (64, 128, (16, 16, 16), 4, 1),     # * m24: Q2 k4x1
    (64, 128, (8, 8, 16), 1, 0),       # Q2 k4x1, no activation
    (128, 128, (16, 16, 16), 4, 1),    # * m27 / m41: Q2 k4x2
    (128, 128, (16, 16, 16), 1, 1),    # batch 1: splits K in two when scratch is offered
    (128, 128, (6, 6, 20), 1, 1),      # ragged
    (64, 128, (8, 8, 8), 1, 1),        # W = 8 bricks (4x4x8), Q2 k4x1
    (128, 256, (8, 8, 8), 4, 1),       # * m31: split-K, 2 slices of 4 chunks
    (256, 256, (8, 8, 8), 4, 1),       # * m34: split-K, 2 slices of 8 chunks
    (256, 256, (8, 8, 8), 1, 2),       # batch 1: more slices
    (128, 256, (6, 6, 12), 1, 1),      # W = 12: partial 8-wide bricks
]
IDS = ["c%d_o%d_%dx%dx%d_n%d_a%d" % (c[0], c[1], *c[2], c[3], c[4]) for c in CASES]


def _inputs(case, seed=0):
    c0, cout, (d, h, w), n, act = case
    rs = np.random.RandomState((hash((c0, cout, d, h, w, n)) + seed) & 0xFFFF)
    x0 = torch.from_numpy(rs.randn(n, c0, d, h, w).astype(np.float32))
    wgt = torch.from_numpy((rs.randn(cout, c0, 3, 3, 3) / np.sqrt(27.0 * c0)).astype(np.float32))
    scale = torch.from_numpy(rs.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = torch.from_numpy((rs.randn(cout) * 0.1).astype(np.float32))
    return x0, wgt, scale, shift


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_ks_matches_cpu(device, case, precision):
    x0, wgt, scale, shift = _inputs(case)
    act = case[4]
    got = run_conv(device, x0, None, wgt, scale, shift, act, precision)
    ref = ref_conv(x0, None, wgt, scale, shift, act, precision)
    assert torch.isfinite(got).all()
    ulp = 2.0 ** -10 if precision == "f16" else 2.0 ** -7          # output stored in the 16-bit type: one rounding of the result
    err = (got.double() - ref.double()).abs()
    tol = ulp * ref.abs().double() + 1e-3 * ulp + 2e-5
    assert (err <= tol).all(), f"max err {err.max().item():.3e} rel_l2 {rel_l2(got, ref):.3e}"


@pytest.mark.parametrize("case", [CASES[9], CASES[12], CASES[13]], ids=[IDS[9], IDS[12], IDS[13]])
def test_ks_without_scratch_matches_with_scratch(device, case):
    """The same layer through the un-split plan (no scratch offered): same values up to the fp32 summation order."""
    x0, wgt, scale, shift = _inputs(case)
    a = run_conv(device, x0, None, wgt, scale, shift, case[4], "f16")
    b = run_conv(device, x0, None, wgt, scale, shift, case[4], "f16", no_scratch=True)
    ref = ref_conv(x0, None, wgt, scale, shift, case[4], "f16")
    assert rel_l2(a, ref) < 5e-4 and rel_l2(b, ref) < 5e-4
    assert (a.double() - b.double()).abs().max().item() <= 2.0 ** -9 * float(ref.abs().max())


@pytest.mark.parametrize("case", [CASES[1], CASES[8], CASES[13]], ids=[IDS[1], IDS[8], IDS[13]])
def test_ks_is_deterministic(device, case):
    """Fixed-order reductions (waves 0..3 in LDS, slices 0..S-1 in the reduce kernel): bit-identical across runs."""
    x0, wgt, scale, shift = _inputs(case)
    a = run_conv(device, x0, None, wgt, scale, shift, case[4], "f16")
    for _ in range(3):
        b = run_conv(device, x0, None, wgt, scale, shift, case[4], "f16")
        assert torch.equal(a, b)
