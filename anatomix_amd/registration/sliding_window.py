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
dealt to ranks in contiguous z-ordered runs; each rank accumulates its own windows.  The ONE exchange
step is a z-slab reduce-scatter: the volume's z axis is cut into ``world`` slabs,
rank s owns slab s, and every rank sends to each owner only the planes of (sum_w*f, sum_w) it
actually touched inside that owner's slab (a rank's windows span ~roi + a few scan intervals of z, so
most (rank, slab) pairs exchange nothing) as direct point-to-point transfers -- on RCCL they run
concurrently over the fully connected xGMI links instead of around a ring.  The owner sums what it
receives and normalises ITS slab only; ``return_slab=True`` leaves the result sharded (what a
throughput-oriented caller wants), the default all-gathers the normalised slabs so that every rank
returns the full tensor, like the single-process call.
"""
from __future__ import annotations

import ctypes
import math
from typing import Callable, List, Sequence, Tuple

import torch
import torch.nn as nn
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


def _fused_ok(predictor, inputs, roi) -> bool:
    """The fused window step (amx_unet_forward_windows) takes a bare Unet on the HIP path whose output conv can accumulate straight
    into the fp32 volume: output_nc a multiple of 16 and <= 32, roi width >= 32 (include/anatomix_amd.h; other shapes leave through
    the export pass, which the generic window loop below handles)."""
    from ..model.network import Unet
    return (isinstance(predictor, Unet) and inputs.is_cuda and inputs.shape[1] == 1 and
            predictor._cfg["output_nc"] % 16 == 0 and predictor._cfg["output_nc"] <= 32 and roi[2] >= 32 and
            predictor.hip_unsupported_reason(inputs[:, :, :1, :1, :1].expand(-1, -1, 2, 2, 2)) is None)


def _is_affine_leaf(m) -> bool:
    if isinstance(m, nn.Identity):
        return True
    if isinstance(m, (nn.Dropout, nn.Dropout3d)):
        return not m.training
    return (isinstance(m, nn.Conv3d) and m.kernel_size == (1, 1, 1) and m.stride == (1, 1, 1) and
            m.padding in ((0, 0, 0), "valid", "same") and m.dilation == (1, 1, 1) and m.groups == 1)


# containers whose forward() is known to be nothing but their children applied in order (checked by name: MONAI is not a dependency)
_COMPOSITION_ONLY = {("monai.networks.blocks.dynunet_block", "UnetOutBlock"), ("monai.networks.blocks.convolutions", "Convolution")}


def _pointwise_affine(module) -> bool:
    """True if `module` is PROVABLY a per-voxel AFFINE map A f + b: 1x1x1 convolutions (stride 1, no padding, any bias), identities and
    eval-mode dropouts, composed only by containers whose forward() is pure composition -- nn.Sequential (and subclasses that do not
    override forward) or the whitelisted MONAI blocks (UnetOutBlock = the head of segmentation_utils.py:113-115, Convolution).  A
    module with its OWN forward() may apply functional nonlinearities (softmax, sigmoid, clamp, argmax) that no child reveals, so it
    is never accepted on the strength of its leaves; a numeric probe of additivity and locality backs the structural check."""
    def structural(m) -> bool:
        kids = list(m.children())
        if not kids:
            return _is_affine_leaf(m)
        pure = (isinstance(m, nn.Sequential) and type(m).forward is nn.Sequential.forward) or \
               (type(m).__module__, type(m).__name__) in _COMPOSITION_ONLY
        return pure and all(structural(k) for k in kids)

    # the verdict is cached on the module for the parameter versions it was formed on: the probe below is five forwards of the head and
    # several host synchronisations, once per sliding_window_inference call otherwise
    sig = tuple((id(p), p._version, p.dtype, p.device) for p in module.parameters())
    cached = module.__dict__.get("_amx_affine_verdict")
    if cached is not None and cached[0] == sig:
        return cached[1]

    def remember(v):
        module.__dict__["_amx_affine_verdict"] = (sig, v)
        return v

    if not structural(module):
        return remember(False)
    convs = [m for m in module.modules() if isinstance(m, nn.Conv3d)]
    if not convs:
        return remember(True)                     # identities only
    # numeric guard: h(a + b) - h(a) - h(b) + h(0) = 0 and a one-voxel perturbation stays in its voxel
    w = convs[0].weight
    with torch.no_grad():
        g = torch.Generator(device="cpu").manual_seed(0)
        a = torch.randn(1, convs[0].in_channels, 2, 2, 3, generator=g).to(w.device, w.dtype)
        b = torch.randn(1, convs[0].in_channels, 2, 2, 3, generator=g).to(w.device, w.dtype)
        try:
            h0, ha, hb, hab = module(torch.zeros_like(a)), module(a), module(b), module(a + b)
            if ha.shape[2:] != a.shape[2:]:
                return remember(False)
            scale = float(hab.abs().max()) + 1e-12
            # additivity within the arithmetic of the head's own dtype (a half-precision 1x1x1 head is affine to ~1e-3, not 1e-4)
            tol = max(1e-4, 16.0 * float(torch.finfo(w.dtype).eps))
            if float((hab.float() - ha.float() - hb.float() + h0.float()).abs().max()) > tol * scale + 1e-6:
                import warnings
                warnings.warn("anatomix_amd.sliding_window_inference: a structurally per-voxel-affine head failed the numeric additivity probe; "
                              "the windows take the generic loop (head inside every window)")
                return remember(False)
            a2 = a.clone()
            a2[..., 0, 0, 0] += 1.0
            diff = (module(a2) - ha).abs()
            diff[..., 0, 0, 0] = 0
            return remember(float(diff.max()) <= 1e-6 * scale + 1e-9)
        except Exception:
            return False


def _split_unet_and_head(predictor, inputs, roi):
    """train_segmentation.py:194-199 validates with predictor = nn.Sequential(Unet, UnetOutBlock).  A per-voxel affine head commutes
    with the window averaging -- sum_w w (A f + b) / sum_w w = A (sum_w w f / sum_w w) + b -- so the windows run through the fused
    path up to the Unet's output and the head is applied once to the normalised volume.  Any other head: generic window loop."""
    if isinstance(predictor, nn.Sequential) and len(predictor) >= 2 and _fused_ok(predictor[0], inputs, roi):
        head = predictor[1] if len(predictor) == 2 else nn.Sequential(*list(predictor.children())[1:])
        if not torch.is_grad_enabled() and _pointwise_affine(head):
            return predictor[0], head
    return None


def sliding_window_inference(inputs: torch.Tensor, roi_size, sw_batch_size: int, predictor: Callable,
                             overlap: float = 0.25, mode: str = "constant", sigma_scale: float = 0.125,
                             padding_mode: str = "constant", cval: float = 0.0, group=None, return_slab: bool = False):
    """inputs [B,C,D,H,W] -> [B,Cout,D,H,W]; same call contract as the reference's use of MONAI.
    ``group``: shard the windows over the ranks of a process group (see the module docstring);
    ``return_slab`` (only with ``group``): return ``(slab [B,Cout,z1-z0,H,W], z0, z1)`` -- this rank's normalised z-slab --
    instead of gathering the full volume on every rank."""
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

    # Sharded: this rank's windows only touch the z-planes [t0, t1) of the volume, so it runs on that sub-volume (a contiguous
    # view: z is the outermost spatial axis) and allocates its partial sums for those planes only -- at 256^3 x 16 channels and
    # 8 ranks ~0.6 GB instead of the full 1.07 GB accumulator per rank.
    touched = []
    for r in range(world):
        run = starts[len(starts) * r // world: len(starts) * (r + 1) // world]
        touched.append((min(s3[0] for s3 in run), max(s3[0] for s3 in run) + roi[0]) if run else (0, 0))
    t0, t1 = touched[rank] if world > 1 else (0, size[0])
    if world > 1:
        if t1 == t0:                  # a rank that received no windows still takes part in the exchange (one plane of zeros)
            t1 = t0 + 1
        inputs = inputs[:, :, t0:t1]
        mine = [(z - t0, y, x) for (z, y, x) in mine]
    lsize = (t1 - t0,) + size[1:]

    cnt = torch.zeros(lsize, dtype=torch.float32, device=inputs.device)
    acc = None
    head = None
    fused = _fused_ok(predictor, inputs, roi)
    if not fused:
        split = _split_unet_and_head(predictor, inputs, roi)
        if split is not None:
            (predictor, head), fused = split, True
    if fused:
        acc = _run_fused(inputs, roi, mine, wmap, predictor, cnt)
    else:
        for b0 in range(0, len(mine) * B, sw_batch_size):
            idx = range(b0, min(b0 + sw_batch_size, len(mine) * B))
            sl = [(i // len(mine), mine[i % len(mine)]) for i in idx]      # batch index outermost
            win = torch.cat([inputs[b:b + 1, :, z:z + roi[0], y:y + roi[1], x:x + roi[2]] for b, (z, y, x) in sl])
            pred = predictor(win)
            if acc is None:
                acc = torch.zeros((B, pred.shape[1]) + lsize, dtype=pred.dtype, device=pred.device)
            for k, (b, (z, y, x)) in enumerate(sl):
                acc[b, :, z:z + roi[0], y:y + roi[1], x:x + roi[2]] += wmap * pred[k]
        for (z, y, x) in mine:
            cnt[z:z + roi[0], y:y + roi[1], x:x + roi[2]] += wmap
        if acc is None:       # no windows: the channel count comes from one probe call
            probe = predictor(torch.zeros((1, inputs.shape[1]) + roi, dtype=inputs.dtype, device=inputs.device))
            acc = torch.zeros((B, probe.shape[1]) + lsize, dtype=probe.dtype, device=probe.device)
    if world > 1:
        slab, slab_cnt, z0, z1 = _exchange_slabs(acc, cnt, t0, size[0], touched, rank, world, group)
        _normalize(slab, slab_cnt)
        if head is not None:
            slab = _apply_head(head, slab)
        if return_slab:
            if need_pad:
                raise ValueError("return_slab needs a volume at least as large as the roi (no padding)")
            return slab, z0, z1
        acc = _gather_slabs(slab, size[0], world, group)
    else:
        _normalize(acc, cnt)
        if head is not None:
            acc = _apply_head(head, acc)
    if need_pad:
        zs, ys, xs = pads[4], pads[2], pads[0]
        acc = acc[:, :, zs:zs + orig[0], ys:ys + orig[1], xs:xs + orig[2]]
    return acc


def _apply_head(head, feat, planes=32):
    """The per-voxel head on the normalised feature volume, a few z-planes at a time (bounded temporaries)."""
    out = None
    for z in range(0, feat.shape[2], planes):
        y = head(feat[:, :, z:z + planes])
        if out is None:
            out = torch.empty((feat.shape[0], y.shape[1]) + tuple(feat.shape[2:]), dtype=y.dtype, device=y.device)
        out[:, :, z:z + planes] = y
    return out


def _normalize(acc, cnt):
    """acc[b, c] /= cnt in place (acc [B,C,d,H,W] contiguous, cnt [d,H,W] contiguous)."""
    if acc.is_cuda and acc.dtype == torch.float32 and acc.is_contiguous() and cnt.is_contiguous() and acc.numel():
        lib = _lib.load()
        st = ctypes.c_void_p(torch.cuda.current_stream(acc.device).cuda_stream)
        vox = cnt.numel()
        with torch.cuda.device(acc.device):
            for b in range(acc.shape[0]):
                _lib.check(lib.amx_sw_normalize(_lib.ptr(acc[b]), _lib.ptr(cnt), acc.shape[1], vox, st))
    else:
        acc /= cnt


def slab_bounds(depth: int, world: int) -> List[int]:
    """z-slab s = [bounds[s], bounds[s + 1])."""
    return [depth * r // world for r in range(world + 1)]


def _exchange_slabs(acc, cnt, t0, depth, touched, rank, world, group):
    """z-slab reduce-scatter by direct point-to-point transfers of the touched planes.  ``acc`` [B,C,d,H,W] / ``cnt`` [d,H,W] hold
    this rank's partial sums for the planes [t0, t0 + d) of a volume of ``depth`` planes.  Returns this rank's summed slab of acc,
    the matching summed slab of cnt, and the slab's z range."""
    import torch.distributed as dist
    B, C, _, H, W = acc.shape
    bounds = slab_bounds(depth, world)
    z0, z1 = bounds[rank], bounds[rank + 1]
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None and group is not dist.group.WORLD else (lambda r: r)
    ops, recvs, keep = [], [], []
    for s in range(world):
        a, b = max(touched[rank][0], bounds[s]), min(touched[rank][1], bounds[s + 1])
        if s != rank and a < b:       # my contribution to slab s: B*C planes-stacks of acc + one of cnt, in one message
            buf = torch.cat([acc[:, :, a - t0:b - t0].reshape(B * C, b - a, H, W), cnt[a - t0:b - t0].unsqueeze(0)])
            keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, peer(s), group))
    for r in range(world):
        a, b = max(touched[r][0], z0), min(touched[r][1], z1)
        if r != rank and a < b:
            buf = torch.empty((B * C + 1, b - a, H, W), dtype=acc.dtype, device=acc.device)
            recvs.append((a, b, buf))
            ops.append(dist.P2POp(dist.irecv, buf, peer(r), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    slab = torch.zeros((B, C, z1 - z0, H, W), dtype=acc.dtype, device=acc.device)
    slab_cnt = torch.zeros((z1 - z0, H, W), dtype=cnt.dtype, device=cnt.device)
    a, b = max(touched[rank][0], z0), min(touched[rank][1], z1)
    if a < b:                         # my own planes inside my slab
        slab[:, :, a - z0:b - z0] = acc[:, :, a - t0:b - t0]
        slab_cnt[a - z0:b - z0] = cnt[a - t0:b - t0]
    for a, b, buf in recvs:           # fixed order (ascending source rank): deterministic sums
        slab[:, :, a - z0:b - z0] += buf[:-1].view(B, C, b - a, H, W)
        slab_cnt[a - z0:b - z0] += buf[-1]
    return slab, slab_cnt, z0, z1


def _gather_slabs(slab, depth, world, group):
    """all_gather of the normalised slabs (padded to the tallest one) -> the full [B,C,D,H,W] tensor on every rank."""
    import torch.distributed as dist
    B, C, _, H, W = slab.shape
    bounds = slab_bounds(depth, world)
    tall = max(bounds[r + 1] - bounds[r] for r in range(world))
    mine = slab if slab.shape[2] == tall else F.pad(slab, (0, 0, 0, 0, 0, tall - slab.shape[2]))
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine.contiguous(), group=group)
    return torch.cat([parts[r][:, :, : bounds[r + 1] - bounds[r]] for r in range(world)], dim=2)


FUSED_WINDOW_BATCH = 4     # windows per amx_unet_forward_windows call (fills the chip on the deep levels)
PIPELINE_WINDOWS = True    # keep two window batches in flight on two HIP streams (ordered accumulation, identical results)


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
        cur = torch.cuda.current_stream(dev)
        st = ctypes.c_void_p(cur.cuda_stream)
        wm = wmap.contiguous()
        ngroups = (len(starts) + k - 1) // k
        # two window batches in flight (amx_unet_forward_windows_pipelined): batch g + 1 runs on the other stream and only its
        # accumulating launches wait for batch g -- same sums in the same order, the underfilled deep levels overlap
        pipelined = PIPELINE_WINDOWS and B * ngroups >= 4 and not torch.cuda.is_current_stream_capturing()
        if pipelined:
            if getattr(model, "_sw_streams", None) is None or model._sw_streams[0].device != dev:
                model._sw_streams = [torch.cuda.Stream(dev) for _ in range(2)]
                model._sw_ws = None
            if model._sw_ws is None or model._sw_ws.numel() < need:
                model._sw_ws = torch.empty(need, dtype=torch.uint8, device=dev)
            lanes = [(model._sw_streams[0], ws), (model._sw_streams[1], model._sw_ws)]
            ready = cur.record_event()
            for s_, _ in lanes:
                s_.wait_event(ready)
        call = 0
        for g0 in range(0, len(starts), k):
            grp = starts[g0:g0 + k]
            offs = (ctypes.c_int * (3 * len(grp)))(*[v for s3 in grp for v in s3])
            for b in range(B):
                if pipelined:
                    s_, w_ = lanes[call & 1]
                    _lib.check(lib.amx_unet_forward_windows_pipelined(
                        model._handle, _lib.ptr(x[b]), size[0], size[1], size[2], len(grp), offs, roi[0], roi[1], roi[2],
                        _lib.ptr(wm), _lib.ptr(acc[b]), _lib.ptr(w_), need, call & 1, ctypes.c_void_p(s_.cuda_stream)))
                    call += 1
                else:
                    _lib.check(lib.amx_unet_forward_windows(model._handle, _lib.ptr(x[b]), size[0], size[1], size[2], len(grp),
                                                            offs, roi[0], roi[1], roi[2], _lib.ptr(wm), _lib.ptr(acc[b]),
                                                            _lib.ptr(ws), need, st))
            for (z, y, xx) in grp:              # the count map is independent of the network: caller's stream
                _lib.check(lib.amx_sw_count(_lib.ptr(cnt), size[0], size[1], size[2], z, y, xx, roi[0], roi[1], roi[2],
                                            _lib.ptr(wm), st))
        if pipelined:
            for s_, _ in lanes:
                cur.wait_stream(s_)
        if model.precision in ("f16", "fp16", "float16", "f16x2", "f16x2mx", "strict_mx") and not torch.cuda.is_current_stream_capturing():
            # the accumulation volume leaves the library here: one synchronising range check per volume (f16 storage only)
            model.check_numerics(synchronize=True)
    return acc
