"""SupPatchNCELoss with the reference's constructor and call contract (supcl_model.py:16-226)."""
import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib


class _SupConHip(torch.autograd.Function):
    """loss, d loss / d features from ONE amx_supcon_loss call (forward and backward are fused in the library)."""

    @staticmethod
    def forward(ctx, feat2d, labels, temperature, rarity, balance, sqrt_mode):
        lib = _lib.load()
        dev = feat2d.device
        x = feat2d.detach().float().contiguous()
        n, c = x.shape
        need_grad = feat2d.requires_grad
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        grad = torch.empty_like(x) if need_grad else None
        with torch.cuda.device(dev):
            nbytes = lib.amx_supcon_scratch_bytes(n, c)
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.amx_supcon_loss(_lib.ptr(x), _lib.ptr(labels), n, c, float(temperature), int(rarity), int(balance),
                                           int(sqrt_mode), _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(scratch), nbytes, st))
        ctx.save_for_backward(grad)
        ctx.in_dtype = feat2d.dtype
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return (grad * gout).to(ctx.in_dtype), None, None, None, None, None


class _SupConBatchHip(torch.autograd.Function):
    """Several losses of one shape from ONE chain of launches (amx_supcon_loss_batch): [nb] losses, gradients kept for the backward."""

    @staticmethod
    def forward(ctx, labels, temperature, rarity, balance, sqrt_mode, *feats):
        lib = _lib.load()
        nb = len(feats)
        dev = feats[0].device
        xs = [f.detach().float().contiguous() for f in feats]
        n, c = xs[0].shape
        need_grad = any(f.requires_grad for f in feats)
        loss = torch.empty(nb, dtype=torch.float32, device=dev)
        grad = torch.empty((nb, n, c), dtype=torch.float32, device=dev) if need_grad else None
        with torch.cuda.device(dev):
            per = lib.amx_supcon_scratch_bytes(n, c)
            scratch = torch.empty(nb * per, dtype=torch.uint8, device=dev)
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
            _lib.check(lib.amx_supcon_loss_batch(nb, arr(xs), arr(list(labels)), n, c, float(temperature), int(rarity), int(balance),
                                                 int(sqrt_mode), arr(list(loss)), None if grad is None else arr(list(grad)),
                                                 _lib.ptr(scratch), nb * per, st))
        ctx.save_for_backward(grad)
        ctx.in_dtypes = [f.dtype for f in feats]
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        g = grad * gout.view(-1, 1, 1)
        return (None, None, None, None, None) + tuple(g[b].to(dt) for b, dt in enumerate(ctx.in_dtypes))


def batched_losses(criterions, features, labels_seg, coords, ranges):
    """[crit(f, labels_seg, c, r) for ...] as ONE tensor [len(criterions)] from one chain of launches, or None when the criteria or
    their inputs are not of one kind (the caller then evaluates them one by one): CUDA fp32 features of one shape, 3-D coordinate
    ranges, int64 [P, 3] device coordinates, one fp32 [1, 1, D, H, W] device segmentation, criteria of equal settings."""
    nb = len(criterions)
    if not (1 <= nb <= 8 and all(type(c) is SupPatchNCELoss for c in criterions)):
        return None
    c0 = criterions[0]
    key = lambda c: (c.temperature, bool(c.weigh_rarity), bool(c.balance_denominator), c.weighting_mode)
    if any(key(c) != key(c0) for c in criterions):
        return None
    f0 = features[0]
    if not (f0.is_cuda and f0.dim() == 3 and all(f.is_cuda and f.device == f0.device and f.shape == f0.shape and f.dtype == torch.float32
                                                   for f in features)):
        return None
    ntps, num_patches, nc = f0.shape
    if not (labels_seg.is_cuda and labels_seg.dtype == torch.float32 and labels_seg.dim() == 5 and labels_seg.shape[:2] == (1, 1) and
            labels_seg.device == f0.device):
        return None
    for c, r in zip(coords, ranges):
        if not (len(r) == 3 and torch.is_tensor(c) and c.dtype == torch.int64 and c.is_cuda and c.device == f0.device and
                tuple(c.shape) == (num_patches, 3)):
            return None
    lib = _lib.load()
    seg = labels_seg.contiguous()
    cs = [c.contiguous() for c in coords]
    lab = torch.empty((nb, ntps * num_patches), dtype=torch.int32, device=f0.device)
    dims = (ctypes.c_int * (3 * nb))(*[int(v) for r in ranges for v in r])
    with torch.cuda.device(f0.device):
        st = ctypes.c_void_p(torch.cuda.current_stream(f0.device).cuda_stream)
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        _lib.check(lib.amx_gather_labels_batch(_lib.ptr(seg), seg.shape[2], seg.shape[3], seg.shape[4], nb, arr(cs), num_patches, dims, ntps,
                                               arr(list(lab)), st))
    return _SupConBatchHip.apply(lab, c0.temperature, c0.weigh_rarity, c0.balance_denominator, c0.weighting_mode == "sqrt",
                                 *[f.reshape(ntps * num_patches, nc) for f in features])


class SupPatchNCELoss(nn.Module):
    """``SupPatchNCELoss(opt)``; ``opt`` carries nce_T, weigh_rarity, balance_denominator, weighting_mode
    (supcl_model.py:49-58).  ``forward(features [views,P,C], labels_seg [1,1,H,W,D], labels_coords [P,3], coords_range)``
    returns the scalar loss.  CUDA features run on the HIP kernel; CPU features raise unless ``allow_torch_path``."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.mask_dtype = torch.bool
        self.temperature = opt.nce_T
        self.weigh_rarity = getattr(opt, "weigh_rarity", False)
        self.balance_denominator = getattr(opt, "balance_denominator", False)
        self.weighting_mode = getattr(opt, "weighting_mode", "raw")
        self.allow_torch_path = os.environ.get("AMX_ALLOW_TORCH_PATH", "0") == "1"

    @staticmethod
    def gather_labels(labels_seg, labels_coords, coords_range):
        """supcl_model.py:100-123: nearest-resize the segmentation to the feature map, gather at the coordinates."""
        n = len(coords_range)
        if n not in (2, 3):
            raise NotImplementedError
        seg = F.interpolate(labels_seg, size=tuple(int(v) for v in coords_range), mode="nearest").squeeze(1)
        if n == 3:
            return seg[:, labels_coords[:, 0], labels_coords[:, 1], labels_coords[:, 2]]
        return seg[:, labels_coords[:, 0], labels_coords[:, 1]]

    @torch.compiler.disable      # the reference compiles its criterion (supcl_model.py:477-489); the kernel call is opaque
    def forward(self, features, labels_seg, labels_coords, coords_range, debug=False):
        ntps, num_patches, nc = features.size()
        # (the kernel takes raw device pointers: coordinates that are not an int64 [P, 3] tensor on the features' device -- a host tensor, a
        #  list, fewer rows than patches -- keep the indexing route below, which raises or converts like the reference's torch ops)
        if (features.is_cuda and len(coords_range) == 3 and labels_seg.is_cuda and labels_seg.dtype == torch.float32 and
                labels_seg.dim() == 5 and labels_seg.shape[:2] == (1, 1) and labels_seg.device == features.device and
                torch.is_tensor(labels_coords) and labels_coords.dtype == torch.int64 and labels_coords.is_cuda and
                labels_coords.device == features.device and tuple(labels_coords.shape) == (num_patches, 3)):
            # the class ids of the views x P patches in one launch (amx_gather_labels) instead of resize + index + round + cast + repeat
            import ctypes
            from .. import _lib
            lib = _lib.load()
            seg = labels_seg.contiguous()
            coords = labels_coords.contiguous()
            lab = torch.empty(ntps * num_patches, dtype=torch.int32, device=features.device)
            with torch.cuda.device(features.device):
                st = ctypes.c_void_p(torch.cuda.current_stream(features.device).cuda_stream)
                _lib.check(lib.amx_gather_labels(_lib.ptr(seg), seg.shape[2], seg.shape[3], seg.shape[4], _lib.ptr(coords), num_patches,
                                                 int(coords_range[0]), int(coords_range[1]), int(coords_range[2]), ntps, _lib.ptr(lab), st))
            return _SupConHip.apply(features.reshape(ntps * num_patches, nc), lab, self.temperature, self.weigh_rarity,
                                    self.balance_denominator, self.weighting_mode == "sqrt")
        labels = self.gather_labels(labels_seg, labels_coords, coords_range)        # [bs = 1, P]
        if labels.shape[0] != 1:
            raise NotImplementedError("one segmentation shared by the views (the reference's eq(labels, labels.T) needs bs == 1)")
        if labels.shape[1] != num_patches:     # (the reference's mask arithmetic fails on the shape mismatch; the kernel would read past the labels)
            raise RuntimeError(f"labels_coords has {labels.shape[1]} rows, features have {num_patches} patches")
        if features.is_cuda:
            # class ids as int32, tiled over the views in (view, patch) order = features.view(ntps * P, nc)
            lab = labels[0].to(device=features.device).round().to(torch.int32).repeat(ntps).contiguous()
            return _SupConHip.apply(features.reshape(ntps * num_patches, nc), lab, self.temperature, self.weigh_rarity,
                                    self.balance_denominator, self.weighting_mode == "sqrt")
        if not self.allow_torch_path:
            raise RuntimeError("anatomix_amd.SupPatchNCELoss: features are not on a GPU; set allow_torch_path = True "
                               "(or AMX_ALLOW_TORCH_PATH=1) to evaluate the loss with stock torch ops")
        return self._forward_torch(features, labels)

    def _forward_torch(self, features, labels):
        ntps, p, nc = features.size()
        x = F.normalize(features.view(ntps * p, nc), dim=-1, eps=1e-8)
        logits = x @ x.t() / self.temperature
        logits = logits - logits.max(dim=1, keepdim=True).values.detach()
        same = torch.eq(labels, labels.T).float().repeat(ntps, ntps)
        counts = same.sum(1)
        off = 1.0 - torch.eye(ntps * p, dtype=same.dtype, device=same.device)
        pos = same * off
        if self.balance_denominator:
            npc = counts.unsqueeze(0) - same
            if self.weighting_mode == "sqrt":
                npc = npc.sqrt()
            log_prob = logits - torch.logsumexp(logits + torch.log(off / npc), dim=1, keepdim=True)
        else:
            log_prob = logits - torch.log((torch.exp(logits) * off).sum(1, keepdim=True))
        loss = -(pos * log_prob).sum(1) / pos.sum(1)
        if self.weigh_rarity:
            w = 1.0 / (counts.sqrt() if self.weighting_mode == "sqrt" else counts)
            return (w * loss).sum() / w.sum()
        return loss.view(ntps, p).mean()
