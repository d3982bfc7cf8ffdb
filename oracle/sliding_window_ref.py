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


def block_probe(y, vol, roi, variant="anatomix", overlap=0.8, sigma_scale=0.25, seed=0):
    """Parity probe of a cubic sliding-window result at full schedule size without 343 CPU forwards: the block of output voxels
    [s1, s2)^3 between the second and the third window start of every axis is covered by exactly the 8 windows with starts in
    {s0, s1}^3, so its oracle value needs 8 forwards of the fp32 restatement (oracle/unet_ref.py, synthetic weights `seed`) blended
    with gaussian_map.  y: [1, C, V, V, V] result (torch, any device), vol: [1, 1, V, V, V] input.  Returns the distances (dict) or
    None when the schedule has fewer than three starts per axis."""
    import os
    import torch
    from oracle import unet_ref as R
    kw = R.VARIANTS[variant]
    sd = R.synthetic_state_dict(kw, seed)
    V, S = vol.shape[-1], roi
    starts = starts_1d(V, S, overlap)
    if len(starts) < 3 or starts[2] - starts[1] < 1:
        return None
    lo, hi = starts[1], starts[2]
    # no other window may touch the block: the third start is its upper bound and every later start lies beyond it
    assert all(st >= hi for st in starts[2:]) and starts[0] + S >= hi and starts[1] + S >= hi
    w = torch.from_numpy(gaussian_map((S, S, S), sigma_scale))
    volc = vol.detach().float().cpu()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(min(avail, 32))
    acc, cnt = None, torch.zeros((hi - lo,) * 3)
    with torch.no_grad():
        for z in starts[:2]:
            for yy in starts[:2]:
                for x in starts[:2]:
                    pred = R.forward(volc[:, :, z:z + S, yy:yy + S, x:x + S], sd, kw)[0]
                    sl = (slice(lo - z, hi - z), slice(lo - yy, hi - yy), slice(lo - x, hi - x))
                    wb = w[sl]
                    acc = wb * pred[(slice(None),) + sl] if acc is None else acc + wb * pred[(slice(None),) + sl]
                    cnt += wb
    ref = (acc / cnt).double()
    got = y.detach()[0, :, lo:hi, lo:hi, lo:hi].double().cpu()
    d = got - ref
    return {"rel_l2_vs_fp32_cpu_oracle": float("%.3e" % float(d.norm() / ref.norm())),
            "max_rel_vs_fp32_cpu_oracle": float("%.3e" % float(d.abs().max() / ref.abs().max())),
            "tolerance": 1e-3, "compliant": bool(float(d.norm() / ref.norm()) <= 1e-3),
            "against": f"numpy sliding-window restatement over the fp32 CPU oracle on the {hi - lo}^3 output block [{lo}, {hi})^3 (covered by "
                       f"exactly 8 of the {len(starts) ** 3} windows: 8 CPU forwards); MONAI itself is absent from the image -- parity with it unpinned"}
