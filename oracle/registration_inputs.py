"""Seeded synthetic inputs of the registration post-processing cases (shared by oracle/make_golden_registration.py and
the tests; test infrastructure only).  Images are smooth random fields in [0, 1] (like min-max-normalised volumes), the
"network features" are random with a smooth component so that the SSD volume has meaningful minima."""
import numpy as np

# case -> (H, W, D, feature channels, MIND radius, dilation, grid_sp, disp_hw)
CASES = {
    "small_r1": (20, 24, 28, 16, 1, 2, 2, 1),
    "small_r2": (16, 20, 24, 16, 2, 2, 2, 2),
    "odd_g4": (24, 16, 40, 8, 1, 2, 4, 1),
}
SCALE = 0.1


def _smooth(rs, shape, passes=2):
    a = rs.rand(*shape).astype(np.float32)
    for _ in range(passes):
        for ax in range(a.ndim - 3, a.ndim):
            a = (a + np.roll(a, 1, ax) + np.roll(a, -1, ax)) / np.float32(3)
    a = (a - a.min()) / (a.max() - a.min())
    return a.astype(np.float32)


def inputs(case):
    h, w, d, c, radius, dilation, g, hw = CASES[case]
    rs = np.random.RandomState(abs(hash_name(case)) % (1 << 31))
    img_f = _smooth(rs, (h, w, d))
    img_m = (0.7 * np.roll(img_f, (1, -1, 2), (0, 1, 2)) + 0.3 * _smooth(rs, (h, w, d))).astype(np.float32)
    feat_f = (_smooth(rs, (c, h, w, d), 1) * 4 + rs.randn(c, h, w, d).astype(np.float32) * 0.2).astype(np.float32)
    feat_m = (np.roll(feat_f, (1, -1, 2), (1, 2, 3)) + rs.randn(c, h, w, d).astype(np.float32) * 0.1).astype(np.float32)
    return img_f, img_m, feat_f, feat_m, radius, dilation, g, hw, SCALE


def hash_name(s):
    v = 0
    for ch in s:
        v = (v * 131 + ord(ch)) & 0x7FFFFFFF
    return v


# masked merge_features cases (instance_optimization.py:52-97): case -> (H, W, D, feature channels); even dims (the reference's
# 2x subsample + trilinear x2 round trip only closes on even dims)
MASK_CASES = {"mask_even": (20, 24, 28, 4), "mask_cube": (16, 16, 16, 2)}


def mask_inputs(case):
    """(img_f, img_m, feat_f, feat_m, mask_f, mask_m): float32 0/1 ellipsoid masks that leave a margin on every side."""
    h, w, d, c = MASK_CASES[case]
    rs = np.random.RandomState(abs(hash_name(case)) % (1 << 31))
    img_f = _smooth(rs, (h, w, d))
    img_m = (0.6 * np.roll(img_f, (1, 2, -1), (0, 1, 2)) + 0.4 * _smooth(rs, (h, w, d))).astype(np.float32)
    feat_f = rs.randn(c, h, w, d).astype(np.float32)
    feat_m = rs.randn(c, h, w, d).astype(np.float32)
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, h), np.linspace(-1, 1, w), np.linspace(-1, 1, d), indexing="ij")
    mask_f = ((zz / 0.8) ** 2 + (yy / 0.7) ** 2 + (xx / 0.75) ** 2 < 1).astype(np.float32)
    mask_m = (((zz - 0.1) / 0.7) ** 2 + (yy / 0.8) ** 2 + ((xx + 0.05) / 0.8) ** 2 < 1).astype(np.float32)
    return img_f, img_m, feat_f, feat_m, mask_f, mask_m
