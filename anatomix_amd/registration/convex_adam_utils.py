"""Feature side of the registration pipeline with the surface of ``anatomix.registration.convex_adam_utils``:
model loading, min-max normalisation and sliding-window feature extraction (reference lines 16-78, 134-221), and the
feature post-processing that runs on the extracted tensors before the convex optimisation -- ``MINDSSC`` (:311-406),
``apply_avg_pool3d`` (:105-131) and the SSD correlation volume ``correlate`` (:409-491) -- on the HIP kernels of
``csrc/amx_regfeat.hip``.  The solver itself (coupled_convex, inverse_consistency, the Adam instance optimisation,
Jacobian utilities) is downstream of the feature path and not part of this package.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

from .. import _lib
from ..model.load_from_hf import ANATOMIX_VARIANTS, _load_handling_compile, load_from_hf  # noqa: F401
from ..model.network import Unet
from .sliding_window import sliding_window_inference


def load_model(ckpt_path=None, hf_variant=None, *, num_downs=4, ngf=16, output_nc=16, norm="batch", interp="nearest",
               pooling="Max", device=None, weights_path=None):
    """convex_adam_utils.py:16-78.  Exactly one of ``ckpt_path`` / ``hf_variant``; the architecture arguments are
    keyword-only and only used with ``ckpt_path`` (a variant brings its own).  ``hf_variant`` goes through
    ``load_from_hf`` (Hub download, or the local ``weights_path=`` extension of this package where there is no network),
    so the returned model ALWAYS carries loaded weights.  Returned in eval mode on ``device`` (default: cuda if present)."""
    if (ckpt_path is None) == (hf_variant is None):
        raise ValueError("Provide exactly one of `ckpt_path` or `hf_variant`.")
    if hf_variant is not None:
        model = load_from_hf(hf_variant, weights_path=weights_path)
    elif ckpt_path == "scratch":
        raise ValueError("'scratch' is not supported for registration; registration requires pretrained weights.")
    else:
        if not os.path.isfile(ckpt_path):
            raise FileNotFoundError(f"Checkpoint file not found: {ckpt_path}")
        model = Unet(3, 1, output_nc, num_downs, ngf=ngf, norm=norm, interp=interp, pooling=pooling)
        model = _load_handling_compile(model, torch.load(ckpt_path, map_location="cpu"))
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    model.to(device)
    model.eval()
    return model


def minmax(arr, minclip=None, maxclip=None):
    """convex_adam_utils.py:134-156.  The reference's condition ``not (minclip is None) & (maxclip is
    None)`` parses as ``not (both None)``: clipping is applied as soon as EITHER bound is given
    (np.clip accepts None for the other).  No zero-range guard, as in the reference."""
    if not ((minclip is None) and (maxclip is None)):
        arr = np.clip(arr, minclip, maxclip)
    return (arr - arr.min()) / (arr.max() - arr.min())


def extract_features(img_fixed, img_moving, model, fixminclip=None, fixmaxclip=None, movminclip=None, movmaxclip=None,
                     group=None):
    """convex_adam_utils.py:159-221: min-max normalise, then 128^3 / overlap 0.8 / gaussian(0.25)
    sliding-window inference of both volumes.  Returns (fixed_features, moving_features), each
    [1, C, D, H, W] on the model's device."""
    dev = next(model.parameters()).device
    outs = []
    for img, lo, hi in ((img_fixed, fixminclip, fixmaxclip), (img_moving, movminclip, movmaxclip)):
        im = torch.from_numpy(np.ascontiguousarray(minmax(img, lo, hi)))[None, None, ...].float().to(dev)
        with torch.no_grad():
            outs.append(sliding_window_inference(im, (128, 128, 128), 2, model, overlap=0.8, mode="gaussian",
                                                 sigma_scale=0.25, group=group))
    return outs[0], outs[1]


# ---- feature post-processing on the HIP kernels (fp32, batch 1, like the reference's tensors) -------------------------

def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _f32c(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: the registration feature kernels run on the GPU (got a {t.device} tensor); there is no CPU path")
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def MINDSSC(img, radius=2, dilation=2):
    """convex_adam_utils.py:311-406.  img [1, 1, H, W, D] -> MIND-SSC descriptor [1, 12, H, W, D] (fp32).  Unlike the
    reference there is no host synchronisation: the global mean used for the variance clamp (:389-393) stays on the GPU."""
    if img.dim() != 5 or img.shape[0] != 1 or img.shape[1] != 1:
        raise ValueError(f"MINDSSC expects [1, 1, H, W, D] (got {tuple(img.shape)})")
    lib = _lib.load()
    x = _f32c(img, "MINDSSC")
    _, _, h, w, d = x.shape
    out = torch.empty((1, 12, h, w, d), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        nb = lib.amx_mindssc_scratch_bytes(h, w, d)
        sc = torch.empty(nb, dtype=torch.uint8, device=x.device)
        _lib.check(lib.amx_mindssc(_lib.ptr(x), h, w, d, int(radius), int(dilation), _lib.ptr(out), _lib.ptr(sc), nb,
                                   _stream(x.device)))
    return out


def apply_avg_pool3d(disp_hr, kernel_size, num_repeats):
    """convex_adam_utils.py:105-131: ``num_repeats`` x F.avg_pool3d(kernel_size, padding=kernel_size // 2, stride=1).
    disp_hr [1, C, H, W, D] (forward only: the instance optimisation's autograd use of this function is outside the
    feature path)."""
    if disp_hr.dim() != 5 or disp_hr.shape[0] != 1:
        raise ValueError(f"apply_avg_pool3d expects [1, C, H, W, D] (got {tuple(disp_hr.shape)})")
    if disp_hr.requires_grad and torch.is_grad_enabled():
        raise RuntimeError("apply_avg_pool3d on the HIP kernels is forward-only")
    lib = _lib.load()
    x = _f32c(disp_hr, "apply_avg_pool3d")
    _, c, h, w, d = x.shape
    with torch.cuda.device(x.device):
        for _ in range(int(num_repeats)):
            y = torch.empty_like(x)
            _lib.check(lib.amx_box_filter3d(_lib.ptr(x), _lib.ptr(y), c, h, w, d, int(kernel_size), _stream(x.device)))
            x = y
    return x


def smooth_merged_features(mind, pred, grid_sp, downscale_feat_scalar=0.1):
    """``F.avg_pool3d(cat([mind, pred * downscale_feat_scalar], 1), grid_sp, stride=grid_sp)`` in one pass
    (run_convex_adam_with_network_feats.py:164-205 + instance_optimization.py:111-117) without materialising the scaled
    copy and the 28-channel concat at full resolution.  mind [1, 12, H, W, D] or None, pred [1, C, H, W, D]."""
    lib = _lib.load()
    b = _f32c(pred, "smooth_merged_features")
    a = None if mind is None else _f32c(mind, "smooth_merged_features")
    _, cb, h, w, d = b.shape
    ca = 0 if a is None else a.shape[1]
    g = int(grid_sp)
    out = torch.empty((1, ca + cb, h // g, w // g, d // g), dtype=torch.float32, device=b.device)
    with torch.cuda.device(b.device):
        _lib.check(lib.amx_avg_pool3d_cat(_lib.ptr(a), ca, 1.0, _lib.ptr(b), cb, float(downscale_feat_scalar), h, w, d, g,
                                          _lib.ptr(out), _stream(b.device)))
    return out


def correlate(mind_fix, mind_mov, disp_hw, grid_sp, shape, ch=12):
    """convex_adam_utils.py:409-491.  mind_fix, mind_mov [1, ch, H/grid_sp, W/grid_sp, D/grid_sp] ->
    (ssd [(2 disp_hw + 1)^3, h, w, d], ssd_argmin int64 [h, w, d])."""
    lib = _lib.load()
    f = _f32c(mind_fix, "correlate")
    m = _f32c(mind_mov, "correlate")
    h, w, d = int(shape[0]) // grid_sp, int(shape[1]) // grid_sp, int(shape[2]) // grid_sp
    if tuple(f.shape) != (1, ch, h, w, d) or tuple(m.shape) != tuple(f.shape):
        raise ValueError(f"correlate: features {tuple(f.shape)} / {tuple(m.shape)} do not match (1, {ch}, {h}, {w}, {d})")
    k = 2 * int(disp_hw) + 1
    ssd = torch.empty((k ** 3, h, w, d), dtype=torch.float32, device=f.device)
    amin = torch.empty((h, w, d), dtype=torch.int64, device=f.device)
    with torch.cuda.device(f.device):
        nb = lib.amx_correlate_scratch_bytes(h, w, d, int(disp_hw))
        sc = torch.empty(nb, dtype=torch.uint8, device=f.device)
        _lib.check(lib.amx_correlate_ssd(_lib.ptr(f), _lib.ptr(m), ch, h, w, d, int(disp_hw), _lib.ptr(ssd), _lib.ptr(amin),
                                         _lib.ptr(sc), nb, _stream(f.device)))
    return ssd, amin


def stage1_inputs(img_fixed, img_moving, model, grid_sp=2, disp_hw=1, downscale_feat_scalar=0.1, fixminclip=None,
                  fixmaxclip=None, movminclip=None, movmaxclip=None, group=None):
    """Everything the reference computes between loading the two volumes and its convex solver
    (run_convex_adam_with_network_feats.py:153-205 -> instance_optimization.run_stage1_registration's ``correlate`` call):
    min-max + sliding-window features of both volumes, MIND-SSC(1, 2) of both, ``cat(mind, 0.1 * features)`` pooled by
    ``grid_sp``, and the SSD correlation volume fixed -> moving.  img_* are numpy volumes [H, W, D].
    Returns a dict(features_fix_smooth, features_mov_smooth, ssd, ssd_argmin, mind_fixed, mind_moving, pred_fixed,
    pred_moving); all device tensors, no host synchronisation after the inputs are uploaded."""
    pred_f, pred_m = extract_features(img_fixed, img_moving, model, fixminclip, fixmaxclip, movminclip, movmaxclip, group=group)
    dev = pred_f.device
    out = {"pred_fixed": pred_f, "pred_moving": pred_m}
    smooth = []
    for name, img, lo, hi, pred in (("fixed", img_fixed, fixminclip, fixmaxclip, pred_f), ("moving", img_moving, movminclip, movmaxclip, pred_m)):
        im = torch.from_numpy(np.ascontiguousarray(minmax(img, lo, hi)))[None, None, ...].float().to(dev)
        mind = MINDSSC(im, 1, 2)                                      # merge_features, instance_optimization.py:107-108
        out["mind_" + name] = mind
        smooth.append(smooth_merged_features(mind, pred, grid_sp, downscale_feat_scalar))
    out["features_fix_smooth"], out["features_mov_smooth"] = smooth
    h, w, d = (int(v) for v in pred_f.shape[-3:])
    out["ssd"], out["ssd_argmin"] = correlate(smooth[0], smooth[1], disp_hw, grid_sp, (h, w, d), smooth[0].shape[1])
    return out
