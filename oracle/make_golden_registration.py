"""Generates tests/golden/registration_golden.npz by running the REFERENCE's registration post-processing functions
on deterministic synthetic inputs.  Run in the build container only:   python oracle/make_golden_registration.py

``anatomix.registration`` cannot be imported here (its __init__ pulls MONAI / nibabel), so this script parses the two
reference files with ``ast``, compiles ONLY the function definitions it needs straight from /root/reference and calls
them -- nothing of the reference is written into this repository; the fixture holds outputs only.  ``.cuda()`` is
shimmed to the identity (generator only) because MINDSSC builds its kernels with ``.cuda()``
(convex_adam_utils.py:357-372).

Inputs are regenerated from seeds by the tests (``inputs(case)`` below is the single definition both sides use).
"""
import ast
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import registration_ref as RR                  # noqa: E402
from oracle.registration_inputs import CASES, MASK_CASES, inputs, mask_inputs      # noqa: E402

REF = "/root/reference/anatomix/registration"


def reference_functions():
    torch.Tensor.cuda = lambda self, *a, **k: self        # noqa: E731  generator-only shim
    nn.Module.cuda = lambda self, *a, **k: self           # noqa: E731  (merge_features builds its pooling module with .cuda())
    from scipy.ndimage import distance_transform_edt
    ns = {"torch": torch, "nn": nn, "F": F, "np": np, "edt": distance_transform_edt}
    wanted = {"convex_adam_utils.py": {"pdist_squared", "MINDSSC", "apply_avg_pool3d", "correlate"},
              "instance_optimization.py": {"merge_features"}}
    for fname, names in wanted.items():
        src = open(os.path.join(REF, fname)).read()
        tree = ast.parse(src)
        body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
        assert {n.name for n in body} == names, (fname, names)
        exec(compile(ast.Module(body=body, type_ignores=[]), os.path.join(REF, fname), "exec"), ns)
    return ns


def probes(rs, arr, n=4096):
    idx = rs.randint(0, arr.size, n).astype(np.int64)
    return idx, arr.reshape(-1)[idx].astype(np.float32)


def masked_merge_golden(ref):
    """merge_features(use_mask=True) (instance_optimization.py:52-97) -> tests/golden/merge_masked_golden.npz: the full
    12-channel MIND-SSC of the mask-filled images and the masked network features, for both volumes."""
    out = {}
    for case in MASK_CASES:
        img_f, img_m, feat_f, feat_m, mask_f, mask_m = mask_inputs(case)
        tf = lambda a: torch.from_numpy(a.copy())          # noqa: E731
        with torch.no_grad():
            mf, mm, cat_f, cat_m = ref["merge_features"](True, tf(feat_f)[None], tf(feat_m)[None], tf(mask_f), tf(mask_m),
                                                         tf(img_f)[None, None], tf(img_m)[None, None])
        assert cat_f.shape[1] == 12 + feat_f.shape[0]
        out[f"{case}|mind_fixed"], out[f"{case}|mind_moving"] = mf[0].numpy(), mm[0].numpy()
        out[f"{case}|pred_fixed"], out[f"{case}|pred_moving"] = cat_f[0, 12:].numpy(), cat_m[0, 12:].numpy()
        print(case, "masked merge_features", tuple(cat_f.shape), "mind mean", float(mf.mean()))
    # an odd extent does not survive the reference's subsample / x2 interpolate round trip: record that it raises
    img = np.random.RandomState(3).rand(10, 12, 13).astype(np.float32)
    msk = np.ones_like(img)
    try:
        ref["merge_features"](True, torch.zeros(1, 2, 10, 12, 13), torch.zeros(1, 2, 10, 12, 13), torch.from_numpy(msk),
                              torch.from_numpy(msk), torch.from_numpy(img)[None, None], torch.from_numpy(img)[None, None])
        out["odd_dim_raises"] = np.array(0)
    except Exception as e:                                  # noqa: BLE001
        print("odd extent:", type(e).__name__, str(e)[:80])
        out["odd_dim_raises"] = np.array(1)
    path = os.path.join(ROOT, "tests", "golden", "merge_masked_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    torch.set_num_threads(8)
    ref = reference_functions()
    out = {}
    worst = 0.0
    for case in CASES:
        img_f, img_m, feat_f, feat_m, radius, dilation, grid_sp, disp_hw, scale = inputs(case)
        tf = lambda a: torch.from_numpy(a)[None, None]    # noqa: E731
        with torch.no_grad():
            mind = ref["MINDSSC"](tf(img_f), radius, dilation)[0].numpy()
        mine = RR.mindssc(img_f, radius, dilation)
        e = float(np.abs(mine - mind).max())
        worst = max(worst, e)
        print(case, "MINDSSC", mind.shape, "oracle-vs-reference max abs", e)
        rs = np.random.RandomState(11)
        out[f"{case}|mind|idx"], out[f"{case}|mind|val"] = probes(rs, mind)
        out[f"{case}|mind|chan_mean"] = mind.mean((1, 2, 3)).astype(np.float64)
        if mind.size <= 1 << 16:
            out[f"{case}|mind|full"] = mind.astype(np.float32)

        # merge_features (no-mask branch) + "* downscale" + avg_pool3d(grid_sp): the smoothed features of both volumes
        with torch.no_grad():
            pf = torch.from_numpy(feat_f)[None] * scale
            pm = torch.from_numpy(feat_m)[None] * scale
            # the reference calls MINDSSC(img, 1, 2) inside merge_features (instance_optimization.py:107-108)
            mf, mm, cat_f, cat_m = ref["merge_features"](False, pf, pm, None, None, tf(img_f), tf(img_m))
            sm_f = F.avg_pool3d(cat_f, grid_sp, stride=grid_sp)[0].numpy()
            sm_m = F.avg_pool3d(cat_m, grid_sp, stride=grid_sp)[0].numpy()
        mine_f = RR.merged_pooled(RR.mindssc(img_f, 1, 2), feat_f, scale, grid_sp)
        e = float(np.abs(mine_f - sm_f).max())
        worst = max(worst, e)
        print(case, "smoothed merged features", sm_f.shape, "oracle-vs-reference", e)
        out[f"{case}|smooth_fix|idx"], out[f"{case}|smooth_fix|val"] = probes(rs, sm_f)
        out[f"{case}|smooth_mov|idx"], out[f"{case}|smooth_mov|val"] = probes(rs, sm_m)

        # correlate on the smoothed features
        h, w, d = img_f.shape
        with torch.no_grad():
            ssd, amin = ref["correlate"](torch.from_numpy(sm_f)[None], torch.from_numpy(sm_m)[None], disp_hw, grid_sp, (h, w, d),
                                         sm_f.shape[0])
        ssd, amin = ssd.numpy(), amin.numpy()
        ssd_mine, amin_mine = RR.correlate(sm_f, sm_m, disp_hw)
        e = float(np.abs(ssd_mine - ssd).max() / np.abs(ssd).max())
        worst = max(worst, e)
        agree = float((amin_mine == amin).mean())
        print(case, "correlate", ssd.shape, "oracle-vs-reference rel", e, "argmin agreement", agree)
        assert agree > 0.999
        out[f"{case}|ssd|idx"], out[f"{case}|ssd|val"] = probes(rs, ssd)
        out[f"{case}|ssd|argmin"] = amin.astype(np.int16)
        out[f"{case}|ssd|disp_mean"] = ssd.mean((1, 2, 3)).astype(np.float64)

        # apply_avg_pool3d on its own (kernel 3 x2 as in correlate, kernel 5 x3 as the final smoothing)
        x = np.random.RandomState(5).randn(3, h // 2, w // 2, d // 2).astype(np.float32)
        for k, rep in ((3, 2), (5, 3)):
            with torch.no_grad():
                y = ref["apply_avg_pool3d"](torch.from_numpy(x)[None], k, rep)[0].numpy()
            e = float(np.abs(RR.box_filter(x, k, rep) - y).max())
            worst = max(worst, e)
            out[f"{case}|box{k}x{rep}|idx"], out[f"{case}|box{k}x{rep}|val"] = probes(rs, y, 1024)
    masked_merge_golden(ref)
    print("worst oracle-vs-reference deviation", worst)
    assert worst < 2e-5
    path = os.path.join(ROOT, "tests", "golden", "registration_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
