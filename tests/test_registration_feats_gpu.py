"""GPU: MIND-SSC, merged + pooled features, box filter and the SSD correlation volume through the C ABI
(csrc/amx_regfeat.hip) against the CPU oracle and the fixtures captured from the reference."""
import os

import numpy as np
import pytest
import torch

from oracle import registration_ref as RR
from oracle.registration_inputs import CASES, inputs

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "registration_golden.npz"))


def dev():
    return torch.device("cuda:0")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def probe(arr, tag):
    return np.asarray(arr, np.float32).reshape(-1)[G[tag + "|idx"]], G[tag + "|val"]


@pytest.mark.parametrize("case", list(CASES))
def test_against_reference_fixtures_and_oracle(case):
    from anatomix_amd.registration import MINDSSC, apply_avg_pool3d, correlate, merge_features, smooth_merged_features
    img_f, img_m, feat_f, feat_m, radius, dilation, g, hw, scale = inputs(case)
    mind = MINDSSC(cu(img_f)[None, None], radius, dilation)[0].cpu().numpy()
    got, want = probe(mind, f"{case}|mind")
    assert np.abs(got - want).max() < 5e-6                                 # exp / division in fp32: a few ulp of 1.0
    assert np.abs(mind - RR.mindssc(img_f, radius, dilation)).max() < 5e-6
    # merged + pooled features, fused and through the reference-shaped merge_features
    m12 = MINDSSC(cu(img_f)[None, None], 1, 2)
    sm_f = smooth_merged_features(m12, cu(feat_f)[None], g, scale)
    sm_m = smooth_merged_features(MINDSSC(cu(img_m)[None, None], 1, 2), cu(feat_m)[None], g, scale)
    for arr, tag in ((sm_f, "smooth_fix"), (sm_m, "smooth_mov")):
        got, want = probe(arr[0].cpu().numpy(), f"{case}|{tag}")
        assert np.abs(got - want).max() < 5e-6
    mf, mm, cat_f, cat_m = merge_features(False, cu(feat_f)[None] * scale, cu(feat_m)[None] * scale, None, None,
                                          cu(img_f)[None, None], cu(img_m)[None, None])
    assert cat_f.shape[1] == 12 + feat_f.shape[0] and torch.equal(cat_f[:, :12], mf) and torch.equal(mf, m12)
    pooled = torch.nn.functional.avg_pool3d(cat_f, g, stride=g)
    assert (pooled - sm_f).abs().max().item() < 2e-6
    # correlation volume
    h, w, d = img_f.shape
    ssd, amin = correlate(sm_f, sm_m, hw, g, (h, w, d), sm_f.shape[1])
    got, want = probe(ssd.cpu().numpy(), f"{case}|ssd")
    assert np.abs(got - want).max() < 1e-5 * np.abs(want).max()
    assert (amin.cpu().numpy() == G[f"{case}|ssd|argmin"]).mean() > 0.999
    assert amin.dtype == torch.int64 and torch.equal(amin, ssd.argmin(0))
    # box filter on its own
    x = np.random.RandomState(5).randn(3, h // 2, w // 2, d // 2).astype(np.float32)
    for k, rep in ((3, 2), (5, 3)):
        y = apply_avg_pool3d(cu(x)[None], k, rep)[0].cpu().numpy()
        got, want = probe(y, f"{case}|box{k}x{rep}")
        assert np.abs(got - want).max() < 2e-6


def test_ragged_sizes_against_oracle():
    from anatomix_amd.registration import MINDSSC, correlate, smooth_merged_features
    rs = np.random.RandomState(3)
    for shape in ((9, 17, 33), (5, 8, 16), (31, 7, 19)):
        img = rs.rand(*shape).astype(np.float32)
        for radius, dil in ((1, 2), (2, 2), (1, 1), (2, 3)):
            got = MINDSSC(cu(img)[None, None], radius, dil)[0].cpu().numpy()
            # white noise: ssd - min(ssd) cancels most of the 125-term box sums, whose summation order differs
            assert np.abs(got - RR.mindssc(img, radius, dil)).max() < 5e-5, (shape, radius, dil)
    feats = rs.randn(5, 13, 10, 21).astype(np.float32)
    for g in (1, 2, 3):
        got = smooth_merged_features(None, cu(feats)[None], g, 0.1)[0].cpu().numpy()
        assert np.abs(got - RR.merged_pooled(np.zeros((0, 13, 10, 21), np.float32), feats, 0.1, g)).max() < 2e-6
    fix = rs.rand(7, 6, 9, 11).astype(np.float32)
    mov = rs.rand(7, 6, 9, 11).astype(np.float32)
    for hw in (1, 2, 3):
        ssd, amin = correlate(cu(fix)[None], cu(mov)[None], hw, 1, (6, 9, 11), 7)
        ref, ref_amin = RR.correlate(fix, mov, hw)
        assert np.abs(ssd.cpu().numpy() - ref).max() < 1e-5 * np.abs(ref).max()
        assert (amin.cpu().numpy() == ref_amin).mean() > 0.995


def test_full_size_properties():
    """Registration sizes (256^3 image, 128^3 grid): properties that need no CPU reference."""
    from anatomix_amd.registration import MINDSSC, correlate, smooth_merged_features
    torch.manual_seed(0)
    img = torch.rand(1, 1, 256, 256, 256, device=dev())
    mind = MINDSSC(img, 1, 2)
    assert mind.shape == (1, 12, 256, 256, 256) and torch.isfinite(mind).all()
    assert mind.min().item() >= 0.0 and mind.max().item() <= 1.0
    assert (mind.amax(1) == 1.0).all()                                     # the best-matching pair has distance 0 -> exp(0)
    shifted = MINDSSC(img * 1.0 + 0.25, 1, 2)                              # invariant to an intensity offset (up to rounding)
    assert (shifted - mind).abs().max().item() < 2e-3
    sub = MINDSSC(img[:, :, :64].contiguous(), 1, 2)                       # stencil locality: far from the cut the values agree
    gm_ratio = (sub[:, :, :56] / mind[:, :, :56]).log().abs().max().item()    # (only the global clamp mean differs: no effect here)
    assert gm_ratio < 1e-4
    feats = torch.randn(1, 16, 256, 256, 256, device=dev())
    sm = smooth_merged_features(mind, feats, 2, 0.1)
    assert sm.shape == (1, 28, 128, 128, 128)
    assert (sm[:, 12:] - torch.nn.functional.avg_pool3d(feats * 0.1, 2, stride=2)).abs().max().item() < 1e-6
    mov = torch.roll(sm, (1, 0, -1), (2, 3, 4))
    ssd, amin = correlate(sm, mov, 1, 2, (256, 256, 256), 28)
    assert ssd.shape == (27, 128, 128, 128)
    want = ((-1 + 1) * 3 + (0 + 1)) * 3 + (1 + 1)
    inner = amin[4:-4, 4:-4, 4:-4]
    assert (inner == want).float().mean().item() == 1.0
    assert ssd[want, 4:-4, 4:-4, 4:-4].abs().max().item() < 1e-6


def test_errors_are_reported():
    from anatomix_amd import _lib
    from anatomix_amd.registration import MINDSSC, apply_avg_pool3d
    with pytest.raises(_lib.AmxError):
        MINDSSC(torch.rand(1, 1, 8, 8, 8, device=dev()), 3, 2)
    with pytest.raises(_lib.AmxError):
        apply_avg_pool3d(torch.rand(1, 2, 8, 8, 8, device=dev()), 4, 1)
    with pytest.raises(RuntimeError):
        MINDSSC(torch.rand(1, 1, 8, 8, 8), 1, 2)                           # CPU tensor: no CPU path


def test_stage1_inputs_composition():
    """extract -> MIND-SSC -> merged pooled features -> correlation volume as one call equals the pieces (128^3 volumes: one
    sliding-window position, so the features are the plain forward)."""
    import sys, os, io, contextlib
    import anatomix_amd
    from anatomix_amd.registration import stage1_inputs
    from oracle import unet_ref as R
    kw = R.VARIANTS["anatomix"]
    with contextlib.redirect_stdout(io.StringIO()):
        model = anatomix_amd.Unet(**kw)
    model.load_state_dict(R.synthetic_state_dict(kw, 0))
    model = model.to(dev()).eval()
    rs = np.random.RandomState(9)
    fixed = rs.rand(128, 128, 128).astype(np.float32) * 3 + 1
    moving = np.roll(fixed, (2, 0, -2), (0, 1, 2)) + rs.rand(128, 128, 128).astype(np.float32) * 0.05
    res = stage1_inputs(fixed, moving, model, grid_sp=2, disp_hw=1)
    assert res["features_fix_smooth"].shape == (1, 28, 64, 64, 64) and res["ssd"].shape == (27, 64, 64, 64)
    norm = (fixed - fixed.min()) / (fixed.max() - fixed.min())
    with torch.no_grad():
        feats = model(cu(norm)[None, None])
    assert (res["pred_fixed"] - feats).abs().max().item() < 2e-3 * feats.abs().max().item()
    mind = RR.mindssc(norm, 1, 2)
    want = RR.merged_pooled(mind, res["pred_fixed"][0].cpu().numpy(), 0.1, 2)
    assert np.abs(res["features_fix_smooth"][0].cpu().numpy() - want).max() < 1e-5
    ssd_ref, amin_ref = RR.correlate(res["features_fix_smooth"][0].cpu().numpy(), res["features_mov_smooth"][0].cpu().numpy(), 1)
    assert np.abs(res["ssd"].cpu().numpy() - ssd_ref).max() < 1e-5 * np.abs(ssd_ref).max()
    assert (res["ssd_argmin"].cpu().numpy() == amin_ref).mean() > 0.999
    # the moving volume is the fixed one shifted by (2, 0, -2) voxels = (1, 0, -1) grid cells: the SSD minimum finds it
    want_idx = ((-1 + 1) * 3 + (0 + 1)) * 3 + (1 + 1)
    inner = res["ssd_argmin"][8:-8, 8:-8, 8:-8]
    assert (inner == want_idx).float().mean().item() > 0.9


@pytest.mark.parametrize("case", ["mask_even", "mask_cube"])
def test_masked_merge_features_matches_reference_fixture(case):
    """merge_features(use_mask=True) (instance_optimization.py:52-97): distance-transform fill outside the eroded mask, MIND-SSC
    of the filled images on the HIP kernel, network features zeroed outside the mask -- against the reference's own outputs
    (oracle/make_golden_registration.py -> tests/golden/merge_masked_golden.npz)."""
    from anatomix_amd.registration import merge_features
    from oracle.registration_inputs import mask_inputs
    gm = np.load(os.path.join(os.path.dirname(__file__), "golden", "merge_masked_golden.npz"))
    img_f, img_m, feat_f, feat_m, mask_f, mask_m = mask_inputs(case)
    mf, mm, cat_f, cat_m = merge_features(True, cu(feat_f)[None], cu(feat_m)[None], cu(mask_f), cu(mask_m),
                                          cu(img_f)[None, None], cu(img_m)[None, None])
    assert cat_f.shape[1] == 12 + feat_f.shape[0] and torch.equal(cat_f[:, :12], mf) and torch.equal(cat_m[:, :12], mm)
    assert np.abs(mf[0].cpu().numpy() - gm[f"{case}|mind_fixed"]).max() < 5e-6
    assert np.abs(mm[0].cpu().numpy() - gm[f"{case}|mind_moving"]).max() < 5e-6
    assert np.array_equal(cat_f[0, 12:].cpu().numpy(), gm[f"{case}|pred_fixed"])
    assert np.array_equal(cat_m[0, 12:].cpu().numpy(), gm[f"{case}|pred_moving"])
