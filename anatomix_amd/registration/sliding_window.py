"""Sliding-window feature extraction: the caller of the UNet hot path in the reference's
registration pipeline (anatomix/registration/convex_adam_utils.py:159-221 ->
``monai.inferers.sliding_window_inference(im, (128,128,128), 2, model, overlap=0.8,
mode="gaussian", sigma_scale=0.25)``).

MONAI is a third-party dependency of the reference (``requirements.txt:12``, unpinned, not vendored
and not installed here), so its algorithm is restated from its published definition
(SURVEY.md Appendix D): constant-pad volumes smaller than the ROI; scan interval
``int(roi*(1-overlap))`` (``roi`` itself when the axis equals the ROI); per axis the first ``d`` with
``d*interval + roi >= size`` gives ``d+1`` windows, start ``min(d*interval, size-roi)``; importance
map = ones or the separable Gaussian clamped below at ``max(min(map), 1e-3)``; result
``sum_w(w*f) / sum_w(w)``.  Parity of this restatement is UNPINNED (no reference test or fixture
holds MONAI outputs); it is pinned by construction-level properties in tests/.

Two execution paths:
  * generic -- any ``predictor`` callable, plain torch ops on the tensors' device (host plumbing);
  * fused   -- when the predictor is an ``anatomix_amd.Unet`` running on the HIP kernels, each window
    is ONE C-ABI call (``amx_unet_forward_window``): the first conv gathers the window straight out
    of the resident volume and the last conv's epilogue does ``acc[slice] += w * features`` in
    place, so neither the window stack nor the per-window predictions are materialised.

Multi-GPU: windows are independent, so with ``group=`` (a torch.distributed process group) they are
dealt to ranks in contiguous z-ordered runs; each rank accumulates its own windows and ONE
``all_reduce(SUM)`` of (sum_w*f, sum_w) precedes the normalisation -- the only exchange step.
"""
from __future__ import annotations

import ctypes
import math
from typing import Callable, List, Sequence, Tuple

import torch
import torch.nn.functional as F

from .. import _lib


def _scan_interval(image_size, roi_size, overlap):
    out = []
    for im, roi in zip(image_size, roi_size):
        if roi == im:
            out.append(int(roi))
        else:
            iv = int(roi * (1.0 - overlap))
            out.append(iv if iv > 0 else 1)
    return tuple(out)


def window_starts(image_size: Sequence[int], roi_size: Sequence[int], overlap: float) -> List[Tuple[int, ...]]:
    """Window corners in MONAI's order (first axis outermost)."""
    interval = _scan_interval(image_size, roi_size, overlap)
    per_axis = []
    for im, roi, iv in zip(image_size, roi_size, interval):
        num = int(math.ceil(float(im) / iv))
        scan = next((d for d in range(num) if d * iv + roi >= im), None)
        count = scan + 1 if scan is not None else 1
        starts = []
        for idx in range(count):
            s = idx * iv
            s -= max(s + roi - im, 0)
            starts.append(s)
        per_axis.append(starts)
    out = [()]
    for starts in per_axis:
        out = [o + (s,) for o in out for s in starts]
    return out


def importance_map(roi_size: Sequence[int], mode: str = "constant", sigma_scale: float = 0.125,
                   device="cpu") -> torch.Tensor:
    """fp32 [D,H,W] window weights (mode 'constant' or 'gaussian')."""
    if mode == "constant":
        return torch.ones(tuple(roi_size), dtype=torch.float32, device=device)
    if mode != "gaussian":
        raise ValueError(f"unsupported mode {mode!r}")
    m = None
    for i, r in enumerate(roi_size):
        sigma = r * sigma_scale
        x = torch.arange(-(r - 1) / 2.0, (r - 1) / 2.0 + 1, dtype=torch.float32, device=device)
        gline = torch.exp(x ** 2 / (-2.0 * sigma ** 2))
        m = gline if m is None else m.unsqueeze(-1) * gline[(None,) * i]
    floor = max(float(m.min().item()), 1e-3)
    return m.clamp_(min=floor).to(torch.float32)


def _fused_ok(predictor, inputs) -> bool:
    from ..model.network import Unet
    return (isinstance(predictor, Unet) and inputs.is_cuda and inputs.shape[1] == 1 and
            predictor.hip_unsupported_reason(inputs[:, :, :1, :1, :1].expand(-1, -1, 2, 2, 2)) is None)


def sliding_window_inference(inputs: torch.Tensor, roi_size, sw_batch_size: int, predictor: Callable,
                             overlap: float = 0.25, mode: str = "constant", sigma_scale: float = 0.125,
                             padding_mode: str = "constant", cval: float = 0.0, group=None) -> torch.Tensor:
    """inputs [B,C,D,H,W] -> [B,Cout,D,H,W]; same call contract as the reference's use of MONAI."""
    if isinstance(roi_size, int):
        roi_size = (roi_size,) * 3
    roi = tuple(int(r) for r in roi_size)
    B = inputs.shape[0]
    orig = tuple(inputs.shape[2:])
    # pad up to the ROI (symmetric, extra voxel at the end), crop back at the end
    pads, need_pad = [], False
    for k in range(2, -1, -1):
        diff = max(roi[k] - orig[k], 0)
        half = diff // 2
        pads += [half, diff - half]
        need_pad |= diff > 0
    if need_pad:
        inputs = F.pad(inputs, pads, mode=padding_mode, value=cval)
    size = tuple(inputs.shape[2:])
    starts = window_starts(size, roi, overlap)
    wmap = importance_map(roi, mode, sigma_scale, inputs.device)

    rank, world = 0, 1
    if group is not None:
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    # contiguous run of windows per rank (z-major order keeps a rank's windows spatially adjacent)
    lo = len(starts) * rank // world
    hi = len(starts) * (rank + 1) // world
    mine = starts[lo:hi]

    cnt = torch.zeros(size, dtype=torch.float32, device=inputs.device)
    acc = None
    if _fused_ok(predictor, inputs):
        acc = _run_fused(inputs, roi, mine, wmap, predictor, cnt)
    else:
        for b0 in range(0, len(mine) * B, sw_batch_size):
            idx = range(b0, min(b0 + sw_batch_size, len(mine) * B))
            sl = [(i // len(mine), mine[i % len(mine)]) for i in idx]      # batch index outermost
            win = torch.cat([inputs[b:b + 1, :, z:z + roi[0], y:y + roi[1], x:x + roi[2]] for b, (z, y, x) in sl])
            pred = predictor(win)
            if acc is None:
                acc = torch.zeros((B, pred.shape[1]) + size, dtype=pred.dtype, device=pred.device)
            for k, (b, (z, y, x)) in enumerate(sl):
                acc[b, :, z:z + roi[0], y:y + roi[1], x:x + roi[2]] += wmap * pred[k]
        for (z, y, x) in mine:
            cnt[z:z + roi[0], y:y + roi[1], x:x + roi[2]] += wmap
        if acc is None:       # a rank that received no windows still takes part in the reduction
            probe = predictor(inputs[:1, :, :roi[0], :roi[1], :roi[2]])
            acc = torch.zeros((B, probe.shape[1]) + size, dtype=probe.dtype, device=probe.device)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    if acc.is_cuda and acc.dtype == torch.float32:
        lib = _lib.load()
        st = ctypes.c_void_p(torch.cuda.current_stream(acc.device).cuda_stream)
        vox = size[0] * size[1] * size[2]
        with torch.cuda.device(acc.device):
            for b in range(B):
                _lib.check(lib.amx_sw_normalize(_lib.ptr(acc[b]), _lib.ptr(cnt), acc.shape[1], vox, st))
    else:
        acc = acc / cnt
    if need_pad:
        zs, ys, xs = pads[4], pads[2], pads[0]
        acc = acc[:, :, zs:zs + orig[0], ys:ys + orig[1], xs:xs + orig[2]]
    return acc


FUSED_WINDOW_BATCH = 4     # windows per amx_unet_forward_windows call (fills the chip on the deep levels)


def _run_fused(inputs, roi, starts, wmap, model, cnt):
    """amx_unet_forward_windows on groups of FUSED_WINDOW_BATCH windows + one amx_sw_count per window, all on
    the current stream (overlapping windows accumulate in issue order)."""
    lib = _lib.load()
    dev = inputs.device
    B = inputs.shape[0]
    size = tuple(inputs.shape[2:])
    x = inputs.detach()
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    cout = model._cfg["output_nc"]
    acc = torch.zeros((B, cout) + size, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        model._ensure_handle(dev)
        if model._weights_stale():
            model._upload_weights(lib, dev)
        k = max(1, min(FUSED_WINDOW_BATCH, len(starts)))
        ws, need = model._get_workspace(lib, k, roi[0], roi[1], roi[2], dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        wm = wmap.contiguous()
        for g0 in range(0, len(starts), k):
            grp = starts[g0:g0 + k]
            offs = (ctypes.c_int * (3 * len(grp)))(*[v for s3 in grp for v in s3])
            for b in range(B):
                _lib.check(lib.amx_unet_forward_windows(model._handle, _lib.ptr(x[b]), size[0], size[1], size[2], len(grp),
                                                        offs, roi[0], roi[1], roi[2], _lib.ptr(wm), _lib.ptr(acc[b]),
                                                        _lib.ptr(ws), need, st))
            for (z, y, xx) in grp:
                _lib.check(lib.amx_sw_count(_lib.ptr(cnt), size[0], size[1], size[2], z, y, xx, roi[0], roi[1], roi[2],
                                            _lib.ptr(wm), st))
    return acc
