"""CPU restatement of the registration feature post-processing that follows feature extraction.

TEST INFRASTRUCTURE ONLY: imported by tests/ and by oracle/make_golden_registration.py, never by the product path.
Pinned by tests/golden/registration_golden.npz, which holds outputs of the reference's own functions
(anatomix/registration/convex_adam_utils.py MINDSSC :311-406, apply_avg_pool3d :105-131, correlate :409-491,
coupled_convex :494-552; instance_optimization.py merge_features :16-119 (no-mask branch); the ``* 0.1`` and
``avg_pool3d(grid_sp)`` of run_convex_adam_with_network_feats.py:164-205) run in the build container.

Everything is written from the formulas, with explicit clamped index gathers instead of the reference's padded
convolutions / unfold, so a transcription slip in either shows up as a mismatch against the fixtures.
"""
import numpy as np

# the six face neighbours of the centre voxel in the reference's order (convex_adam_utils.py:333-340), as offsets
SIX = np.array([[-1, 0, 0], [0, 0, -1], [0, -1, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0]], dtype=np.int64)
# output channel k shows pair PERM[k] (convex_adam_utils.py:395-402)
PERM = np.array([6, 8, 1, 11, 2, 10, 0, 7, 9, 4, 5, 3], dtype=np.int64)


def mind_pairs():
    """The 12 (i, j) index pairs into SIX with i > j and squared distance 2, in row-major (i, j) order
    (convex_adam_utils.py:343-355): channel c compares the samples at SIX[i] and SIX[j]."""
    return [(i, j) for i in range(6) for j in range(6) if i > j and int(((SIX[i] - SIX[j]) ** 2).sum()) == 2]


def _clamped(vol, dz, dy, dx):
    """vol[clamp(z+dz), clamp(y+dy), clamp(x+dx)] for every voxel (replication padding)."""
    h, w, d = vol.shape
    z = np.clip(np.arange(h) + dz, 0, h - 1)
    y = np.clip(np.arange(w) + dy, 0, w - 1)
    x = np.clip(np.arange(d) + dx, 0, d - 1)
    return vol[np.ix_(z, y, x)]


def mindssc(img, radius=2, dilation=2):
    """img float32 [H, W, D] -> MIND-SSC descriptor float32 [12, H, W, D] (convex_adam_utils.py:311-406)."""
    img = np.asarray(img, dtype=np.float32)
    k = 2 * radius + 1
    ssd = []
    for i, j in mind_pairs():
        a, b = SIX[i] * dilation, SIX[j] * dilation
        diff2 = (_clamped(img, *a) - _clamped(img, *b)) ** 2                     # sampled with replication at the border
        acc = np.zeros_like(diff2)
        for tz in range(-radius, radius + 1):
            for ty in range(-radius, radius + 1):
                for tx in range(-radius, radius + 1):
                    acc += _clamped(diff2, tz, ty, tx)                             # box filter, replication again
        ssd.append(acc / np.float32(k ** 3))
    ssd = np.stack(ssd)
    mind = ssd - ssd.min(0, keepdims=True)
    var = mind.mean(0, keepdims=True, dtype=np.float32)
    gm = np.float32(var.mean(dtype=np.float64))
    var = np.clip(var, gm * np.float32(0.001), gm * np.float32(1000.0))
    mind = np.exp(-(mind / var))
    return mind[PERM].astype(np.float32)


def box_filter(x, k, repeats=1):
    """apply_avg_pool3d (convex_adam_utils.py:105-131): avg_pool3d(k, stride 1, padding k//2) -- zero padding counted in
    the divisor -- ``repeats`` times.  x [C, H, W, D]."""
    x = np.asarray(x, dtype=np.float32)
    r = k // 2
    for _ in range(repeats):
        c, h, w, d = x.shape
        p = np.zeros((c, h + 2 * r, w + 2 * r, d + 2 * r), dtype=np.float32)
        p[:, r:r + h, r:r + w, r:r + d] = x
        acc = np.zeros_like(x)
        for tz in range(k):
            for ty in range(k):
                for tx in range(k):
                    acc += p[:, tz:tz + h, ty:ty + w, tx:tx + d]
        x = acc / np.float32(k ** 3)
    return x


def merged_pooled(mind, feats, scale, g):
    """avg_pool3d(cat(mind, scale * feats), g, stride g)  (run_convex_adam_with_network_feats.py:164-205 with
    merge_features' concat, instance_optimization.py:111-117).  mind [12, H, W, D], feats [C, H, W, D]."""
    cat = np.concatenate([np.asarray(mind, np.float32), np.asarray(feats, np.float32) * np.float32(scale)], 0)
    c, h, w, d = cat.shape
    v = cat[:, :h // g * g, :w // g * g, :d // g * g].reshape(c, h // g, g, w // g, g, d // g, g)
    return v.mean((2, 4, 6), dtype=np.float32)


def correlate(fix, mov, disp_hw):
    """SSD correlation volume (convex_adam_utils.py:409-491).  fix, mov [C, h, w, d] (already on the coarse grid).
    Returns (ssd [(2*disp_hw+1)^3, h, w, d] float32, argmin int64 [h, w, d]); displacement index
    m = (dx * k + dy) * k + dz with dz along h, dy along w, dx along d -- the order the reference's view / transpose /
    reshape sequence (:475-488) leaves behind."""
    fix = np.asarray(fix, np.float32)
    mov = np.asarray(mov, np.float32)
    c, h, w, d = fix.shape
    k = 2 * disp_hw + 1
    pad = np.zeros((c, h + 2 * disp_hw, w + 2 * disp_hw, d + 2 * disp_hw), np.float32)
    pad[:, disp_hw:disp_hw + h, disp_hw:disp_hw + w, disp_hw:disp_hw + d] = mov
    ssd = np.zeros((k ** 3, h, w, d), np.float32)
    for dz in range(k):
        for dy in range(k):
            for dx in range(k):
                diff = fix - pad[:, dz:dz + h, dy:dy + w, dx:dx + d]
                ssd[(dx * k + dy) * k + dz] = (diff * diff).sum(0, dtype=np.float32)
    ssd = box_filter(ssd, 3, 2)
    return ssd, ssd.argmin(0).astype(np.int64)
