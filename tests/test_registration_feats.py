"""CPU: the registration post-processing oracle (oracle/registration_ref.py) against the fixtures captured from the
reference's own functions (oracle/make_golden_registration.py -> tests/golden/registration_golden.npz)."""
import os

import numpy as np
import pytest

from oracle import registration_ref as RR
from oracle.registration_inputs import CASES, inputs

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "registration_golden.npz"))


def probe(arr, tag):
    return np.asarray(arr, np.float32).reshape(-1)[G[tag + "|idx"]], G[tag + "|val"]


def test_mind_pairs_are_the_twelve_octahedron_edges():
    pairs = RR.mind_pairs()
    assert len(pairs) == 12 and len(set(pairs)) == 12
    assert sorted(RR.PERM.tolist()) == list(range(12))


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_matches_reference_fixtures(case):
    img_f, img_m, feat_f, feat_m, radius, dilation, g, hw, scale = inputs(case)
    mind = RR.mindssc(img_f, radius, dilation)
    got, want = probe(mind, f"{case}|mind")
    assert np.abs(got - want).max() < 2e-6
    assert np.abs(mind.mean((1, 2, 3)) - G[f"{case}|mind|chan_mean"]).max() < 1e-6
    if f"{case}|mind|full" in G:
        assert np.abs(mind - G[f"{case}|mind|full"]).max() < 2e-6
    sm_f = RR.merged_pooled(RR.mindssc(img_f, 1, 2), feat_f, scale, g)
    sm_m = RR.merged_pooled(RR.mindssc(img_m, 1, 2), feat_m, scale, g)
    for arr, tag in ((sm_f, "smooth_fix"), (sm_m, "smooth_mov")):
        got, want = probe(arr, f"{case}|{tag}")
        assert np.abs(got - want).max() < 2e-6
    ssd, amin = RR.correlate(sm_f, sm_m, hw)
    got, want = probe(ssd, f"{case}|ssd")
    assert np.abs(got - want).max() < 1e-5 * np.abs(want).max()
    assert (amin == G[f"{case}|ssd|argmin"]).mean() > 0.999
    h, w, d = img_f.shape
    x = np.random.RandomState(5).randn(3, h // 2, w // 2, d // 2).astype(np.float32)
    for k, rep in ((3, 2), (5, 3)):
        got, want = probe(RR.box_filter(x, k, rep), f"{case}|box{k}x{rep}")
        assert np.abs(got - want).max() < 2e-6


def test_correlate_finds_a_pure_shift():
    rs = np.random.RandomState(0)
    fix = rs.rand(6, 12, 12, 12).astype(np.float32)
    mov = np.roll(fix, (1, 0, -1), (1, 2, 3))             # mov[p + (1, 0, -1)] == fix[p]
    ssd, amin = RR.correlate(fix, mov, 1)
    k = 3
    want = ((-1 + 1) * k + (0 + 1)) * k + (1 + 1)         # (dx * k + dy) * k + dz with dz = +1, dy = 0, dx = -1
    assert (amin[3:-3, 3:-3, 3:-3] == want).all()


# ---- merge_features(use_mask=True): the distance-transform fill is host logic of the mirror (anatomix_amd/registration/
#      instance_optimization.py); MIND-SSC of the filled image through the oracle must reproduce the reference's descriptors
GM = np.load(os.path.join(os.path.dirname(__file__), "golden", "merge_masked_golden.npz"))


@pytest.mark.parametrize("case", ["mask_even", "mask_cube"])
def test_masked_fill_matches_reference_fixture(case):
    import torch
    from anatomix_amd.registration.instance_optimization import _fill_outside_mask
    from oracle.registration_inputs import mask_inputs
    img_f, img_m, feat_f, feat_m, mask_f, mask_m = mask_inputs(case)
    for img, mask, tag in ((img_f, mask_f, "fixed"), (img_m, mask_m, "moving")):
        filled = _fill_outside_mask(torch.from_numpy(img)[None, None], torch.from_numpy(mask))
        assert filled.shape == (1, 1) + img.shape
        mind = RR.mindssc(filled[0, 0].numpy(), 1, 2)
        assert np.abs(mind - GM[f"{case}|mind_{tag}"]).max() < 5e-6
    assert int(GM["odd_dim_raises"]) == 1


def test_masked_fill_refuses_odd_extents_like_the_reference():
    import torch
    from anatomix_amd.registration.instance_optimization import _fill_outside_mask
    img = torch.rand(1, 1, 10, 12, 13)
    with pytest.raises(IndexError):
        _fill_outside_mask(img, torch.ones(10, 12, 13))
