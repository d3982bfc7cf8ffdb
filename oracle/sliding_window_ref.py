"""Oracle: plain numpy restatement of ``monai.inferers.sliding_window_inference`` as called by the
reference (anatomix/registration/convex_adam_utils.py:202-219).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: MONAI is a third-party dependency of the reference (requirements.txt:12, unpinned),
it is neither vendored under /root/reference nor installed here, and the reference holds no test or
fixture with its outputs.  The algorithm is restated from MONAI's published definition (SURVEY.md
Appendix D); tests pin it by construction-level properties and against the window counts the
reference configuration implies (343 windows for 256^3 / roi 128 / overlap 0.8).
"""
import math

import numpy as np


def starts_1d(size, roi, overlap):
    interval = roi if roi == size else max(int(roi * (1.0 - overlap)), 1)
    num = int(math.ceil(float(size) / interval))
    scan = None
    for d in range(num):
        if d * interval + roi >= size:
            scan = d
            break
    count = scan + 1 if scan is not None else 1
    return [min(i * interval, size - roi) for i in range(count)]


def gaussian_map(roi, sigma_scale):
    m = None
    for i, r in enumerate(roi):
        x = np.arange(-(r - 1) / 2.0, (r - 1) / 2.0 + 1, dtype=np.float32)
        gl = np.exp(x ** 2 / (-2.0 * (r * sigma_scale) ** 2)).astype(np.float32)
        m = gl if m is None else m[..., None] * gl[(None,) * i]
    return np.maximum(m, max(float(m.min()), 1e-3)).astype(np.float32)


def sliding_window(vol, roi, predictor, overlap, mode="constant", sigma_scale=0.125):
    """vol [C,D,H,W] float32 numpy (each axis >= roi); predictor maps [1,C,*roi] -> [1,Cout,*roi]."""
    size = vol.shape[1:]
    w = np.ones(roi, np.float32) if mode == "constant" else gaussian_map(roi, sigma_scale)
    acc, cnt = None, np.zeros(size, np.float32)
    for z in starts_1d(size[0], roi[0], overlap):
        for y in starts_1d(size[1], roi[1], overlap):
            for x in starts_1d(size[2], roi[2], overlap):
                pred = predictor(vol[None, :, z:z + roi[0], y:y + roi[1], x:x + roi[2]])[0]
                if acc is None:
                    acc = np.zeros((pred.shape[0],) + tuple(size), np.float32)
                acc[:, z:z + roi[0], y:y + roi[1], x:x + roi[2]] += w * pred
                cnt[z:z + roi[0], y:y + roi[1], x:x + roi[2]] += w
    return acc / cnt
