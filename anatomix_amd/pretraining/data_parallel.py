"""Plain data parallelism for the contrastive step: one pair of views per rank, replicas of netG + netF, gradients averaged
over the ranks before the optimizer steps (the reference itself is single-GPU and reaches larger
batches through ``--grad_accum_iters``, supcl_model.py:618-661 -- N ranks x 1 pair is the same average as N accumulated
pairs, each pair with its own BatchNorm batch statistics).

``GradientBuckets`` keeps every parameter's ``.grad`` as a VIEW into a few flat fp32 buffers:
  * no pack / unpack passes around the collective (the all-reduce runs on the memory autograd accumulated into, and the
    optimizer reads the reduced values through the same views);
  * a handful of large messages instead of ~130 small ones: xGMI is point-to-point, a ring step is bound by one link
    (~153 GB/s), so the 27.4 MB of the 6M UNet + the MLP heads go out as <= ``bucket_mb``-sized buckets, last layers first;
  * buckets are reduced asynchronously (``async_op=True``) and waited for together; in eager mode
    (``overlap=True``) a bucket is launched from a post-accumulate hook as soon as its last gradient has landed, so the
    projection heads' bucket travels while the UNet backward is still running.
Backend-agnostic (RCCL on the GPUs, gloo in the CPU tests): SUM all-reduce + one in-place scale.
"""
from __future__ import annotations

from typing import Iterable, List

import torch


class GradientBuckets:
    def __init__(self, nets: Iterable[torch.nn.Module], group=None, bucket_mb: float = 16.0, overlap: bool = False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets: List[torch.Tensor] = []
        self._pending = []
        self._handles = []
        self._remaining = []
        limit = max(int(bucket_mb * (1 << 20)) // 4, 1)
        for net in nets:
            params = [p for p in net.parameters() if p.requires_grad]
            if not params:
                continue
            # reverse order: the gradients of the last layers exist first
            groups, cur, cur_n = [], [], 0
            for p in reversed(params):
                if cur and cur_n + p.numel() > limit:
                    groups.append(cur)
                    cur, cur_n = [], 0
                cur.append(p)
                cur_n += p.numel()
            if cur:
                groups.append(cur)
            for grp in groups:
                flat = torch.zeros(sum(p.numel() for p in grp), dtype=torch.float32, device=grp[0].device)
                o = 0
                for p in grp:
                    if p.dtype != torch.float32:
                        raise TypeError("GradientBuckets expects fp32 parameters (the reference keeps fp32 master weights under autocast)")
                    p.grad = flat[o:o + p.numel()].view_as(p)
                    o += p.numel()
                b = len(self.buckets)
                self.buckets.append(flat)
                if overlap and self.world > 1:
                    remaining = {"n": len(grp), "total": len(grp)}
                    self._remaining.append(remaining)

                    def hook(_p, b=b, remaining=remaining):
                        remaining["n"] -= 1
                        if remaining["n"] == 0:
                            remaining["n"] = remaining["total"]
                            self._launch(b)
                    for p in grp:
                        p.register_post_accumulate_grad_hook(hook)
        self.overlap = overlap and self.world > 1

    @property
    def nbytes(self):
        return sum(b.numel() * 4 for b in self.buckets)

    def zero(self):
        """Replaces optimizer.zero_grad(): the views stay, the buffers are cleared (graph-capturable)."""
        for b in self.buckets:
            b.zero_()

    def _launch(self, b):
        self._pending.append(b)
        self._handles.append(self.dist.all_reduce(self.buckets[b], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def sync(self):
        """Average every bucket over the ranks.  Call between backward and the optimizer steps."""
        if self.world == 1:
            return
        launched = set(self._pending)
        for b in range(len(self.buckets)):
            if b not in launched:
                self._launch(b)
        for h in self._handles:
            h.wait()
        inv = 1.0 / self.world
        for b in self.buckets:
            b.mul_(inv)
        self._pending, self._handles = [], []
        for r in self._remaining:          # a parameter that received no gradient this step must not skew the next count
            r["n"] = r["total"]
