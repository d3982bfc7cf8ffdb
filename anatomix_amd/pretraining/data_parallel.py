"""Plain data parallelism for the contrastive step: one pair of views per rank, replicas of netG + netF, gradients averaged
over the ranks before the optimizer steps (the reference itself is single-GPU and reaches larger
batches through ``--grad_accum_iters``, supcl_model.py:618-661 -- N ranks x 1 pair is the same average as N accumulated
pairs, each pair with its own BatchNorm batch statistics).

``GradientBuckets`` owns a few flat fp32 buffers with one VIEW per parameter:
  * after the backward, ``collect()`` copies every gradient into its view (one multi-tensor copy per bucket) and points
    ``p.grad`` at the view, so the collective runs on the flat memory and the optimizer reads the reduced values through the
    same views -- no unpack pass;
  * a handful of large messages instead of ~130 small ones: xGMI is point-to-point, a ring step is bound by one link
    (~153 GB/s), so the 27.4 MB of the 6M UNet + the MLP heads go out as <= ``bucket_mb``-sized buckets, last layers first;
  * buckets are reduced asynchronously (``async_op=True``) and waited for together; in eager mode (``overlap=True``) a bucket
    is collected and launched from a post-accumulate hook as soon as its last gradient exists, so the projection heads'
    bucket travels while the UNet backward is still running;
  * ``release()`` (instead of ``optimizer.zero_grad()``) detaches the parameters from the views again, so that the next
    backward hands autograd fresh gradients.  (Letting autograd accumulate IN PLACE into the views is not graph-safe: an
    AccumulateGrad node outlives the iteration that made it and adds on the stream it was created on, outside a capture --
    measured: HSA memory-aperture fault at the first replay.)
Backend-agnostic (RCCL on the GPUs, gloo in the CPU tests): SUM all-reduce + one in-place scale.
"""
from __future__ import annotations

from typing import Iterable, List

import torch


class GradientBuckets:
    """Replica policy: at construction every parameter AND buffer is broadcast from the group's first rank (``broadcast=True``),
    so the replicas start identical whatever each rank's seed was.  Parameters then stay in step because every rank applies the
    same averaged gradient.  BatchNorm running statistics are per-rank afterwards (each rank sees its own pair of views; the
    reference updates them sequentially pair by pair, supcl_model.py:618-661) -- they only matter for eval-mode use, for which
    rank 0's are the ones a checkpoint would hold; ``sync_buffers()`` re-broadcasts them on demand."""

    def __init__(self, nets: Iterable[torch.nn.Module], group=None, bucket_mb: float = 16.0, overlap: bool = False,
                 broadcast: bool = True):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        nets = list(nets)
        self.nets = nets
        if broadcast and self.world > 1:
            self._broadcast([t for net in nets for t in list(net.parameters()) + list(net.buffers())])
        self.buckets: List[torch.Tensor] = []
        self.params: List[List[torch.nn.Parameter]] = []      # per bucket
        self.views: List[List[torch.Tensor]] = []
        self._launched = set()
        self._collected = set()
        self._handles = []
        self._remaining = []
        self.overlap = bool(overlap) and self.world > 1
        limit = max(int(bucket_mb * (1 << 20)) // 4, 1)
        for net in nets:
            params = [p for p in net.parameters() if p.requires_grad]
            # reverse order: the gradients of the last layers exist first
            groups, cur, cur_n = [], [], 0
            for p in reversed(params):
                if p.dtype != torch.float32:
                    raise TypeError("GradientBuckets expects fp32 parameters (the reference keeps fp32 master weights under autocast)")
                if cur and cur_n + p.numel() > limit:
                    groups.append(cur)
                    cur, cur_n = [], 0
                cur.append(p)
                cur_n += p.numel()
            if cur:
                groups.append(cur)
            for grp in groups:
                flat = torch.zeros(sum(p.numel() for p in grp), dtype=torch.float32, device=grp[0].device)
                views, o = [], 0
                for p in grp:
                    views.append(flat[o:o + p.numel()].view_as(p))
                    o += p.numel()
                b = len(self.buckets)
                self.buckets.append(flat)
                self.params.append(grp)
                self.views.append(views)
                if self.overlap:
                    remaining = {"n": len(grp), "total": len(grp)}
                    self._remaining.append(remaining)

                    def hook(_p, b=b, remaining=remaining):
                        remaining["n"] -= 1
                        if remaining["n"] == 0:
                            remaining["n"] = remaining["total"]
                            self._collect_bucket(b)
                            self._launch(b)
                    for p in grp:
                        p.register_post_accumulate_grad_hook(hook)

    def _broadcast(self, tensors):
        src = self.dist.get_global_rank(self.group, 0) if self.group is not None and self.group is not self.dist.group.WORLD else 0
        with torch.no_grad():
            for t in tensors:
                self.dist.broadcast(t.data, src=src, group=self.group)

    def sync_buffers(self):
        """Re-broadcast every buffer (BatchNorm running statistics, counters) from the group's first rank."""
        if self.world > 1:
            self._broadcast([b for net in self.nets for b in net.buffers()])

    @property
    def nbytes(self):
        return sum(b.numel() * 4 for b in self.buckets)

    def _collect_bucket(self, b):
        if b in self._collected:
            return
        self._collected.add(b)
        dst, src = [], []
        for p, v in zip(self.params[b], self.views[b]):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.params[b], self.views[b]):
            p.grad = v

    def collect(self):
        """Gradients -> flat buckets (one multi-tensor copy per bucket); ``p.grad`` becomes the bucket view.  Graph-capturable."""
        for b in range(len(self.buckets)):
            self._collect_bucket(b)

    def _launch(self, b):
        if b in self._launched or self.world == 1:
            return
        self._launched.add(b)
        self._handles.append(self.dist.all_reduce(self.buckets[b], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def sync(self):
        """Average every bucket over the ranks.  Call between backward (+ collect) and the optimizer steps."""
        self.collect()
        if self.world > 1:
            for b in range(len(self.buckets)):
                self._launch(b)
            for h in self._handles:
                h.wait()
            inv = 1.0 / self.world
            for b in self.buckets:
                b.mul_(inv)
        self._launched, self._handles = set(), []
        for r in self._remaining:          # a parameter that received no gradient this step must not skew the next count
            r["n"] = r["total"]

    def release(self):
        """Replaces ``optimizer.zero_grad()``: the next backward produces fresh gradients (``p.grad = None``)."""
        self._collected = set()
        for grp in self.params:
            for p in grp:
                p.grad = None
