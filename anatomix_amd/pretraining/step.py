"""One contrastive-pretraining step: the per-batch body of the reference's SupCLModel
(forward supcl_model.py:723-770, calculate_NCE_loss 801-843, optimize_parameters 603-661)."""
from collections import OrderedDict
from collections.abc import Mapping

import contextlib
import gc
import weakref
import os

import torch
import torch.nn as nn

from .. import _lib

_PARALLEL_HEADS = _lib.exp_env("AMX_SERIAL_HEADS", "0") != "1"
_SAMPLED_TAPS = _lib.exp_env("AMX_DENSE_TAPS", "0") != "1"      # 0: the dense-tap route (A/B; same values)
_STREAMS = {}
# Distinct side streams for the six per-layer head / loss chains.  One per layer (the first form) is NOT the fastest: HIP maps streams
# onto four hardware queues and every cross-queue dependency of a replayed graph costs -- three streams (two layers each) measured
# 7.45-7.50 ms per step against 7.74-7.77 with six, 7.53-7.55 with two, 7.68 with four (GPU_MAX_HW_QUEUES=8 instead: 14.4 ms).
_HEAD_STREAMS = int(_lib.exp_env("AMX_HEAD_STREAMS", "3"))
_PREDRAW = _lib.exp_env("AMX_NO_PREDRAW", "0") != "1"     # A/B: coordinates drawn up front on a side stream
# A/B: the coordinate draws are enqueued behind the forward's first block instead of in front of it.  (Measured and dropped: the head +
# loss chains started at their taps, beside the rest of the forward, on detached leaves of the rows with a second backward call for the
# network -- 7.04 against 6.86 ms per step whether forked tap by tap or once in front of the 128^3 level: the forward's own chain changes
# hardware queue at every fork of the replayed graph and its short deep-level kernels queue behind the heads'.)
_LATE_DRAW = _lib.exp_env("AMX_DRAW_FIRST", "0") != "1"
_BATCHED_HEADS = _lib.exp_env("AMX_HEAD_CHAINS", "0") != "1"   # A/B: 1 = the six per-layer chains on side streams (until round 6)
_WEIGHTS = {}                                                # (device, nce weights, lambda, accumulation) -> weight vector on the device


def _draw_stream(device):
    """The stream of the up-front coordinate draws: the first head stream (idle until the heads start; a stream of its own measured
    the same, 7.38 vs 7.36 ms per step, and is one more stream on four hardware queues)."""
    return _layer_streams(device, 1)[0]


def _layer_streams(device, n):
    """The side streams of the n per-layer chains (``_HEAD_STREAMS`` distinct ones, shared round-robin), created once per device."""
    key = (device.type, device.index)
    have = _STREAMS.setdefault(key, [])
    nuniq = min(n, _HEAD_STREAMS) if _HEAD_STREAMS > 0 else n
    while len(have) < nuniq:
        have.append(torch.cuda.Stream(device=device))
    return [have[k % nuniq] for k in range(n)]



def _forward_backward(netG, netF, criterions, real_A, real_B, seg_A, nce_layers, nce_weights, num_patches, lambda_nce,
                      sample_ids, grad_accum_iters, scaler=None):
    """forward with taps -> sampler + heads -> per-layer losses -> backward.  Nothing here synchronises with the host."""
    if nce_weights is None:
        nce_weights = [1.0 / len(nce_layers)] * len(nce_layers)
    reals = torch.cat((real_A, real_B), dim=0) if real_B is not None else real_A
    sampled = _sampled_route(netG, netF, reals, nce_layers, num_patches)
    if sampled is not None:
        # netF only ever reads num_patches voxels of each tapped feature map (pretraining_networks.py:472-480): the network hands
        # back those rows -- drawn with netF's own sampler when the forward reaches each tap, i.e. in netF's order, so the generator
        # is consumed exactly as by netG(...) followed by netF(...) -- instead of dense fp32 copies of six tensors
        from ..model import train as _train
        # The coordinates do not depend on the features: once the tap shapes of this (network, input shape) are known from an earlier
        # step they are all drawn up front, in netF's layer order (the generator is consumed exactly as before), on a side stream the
        # forward joins at its first tap -- six draw + filter launches leave the main stream's critical path.
        # (the cache lives ON the module -- a dict keyed by id(netG) would hand a recycled id the plan of a dead network)
        skey = (tuple(reals.shape), tuple(int(l) for l in nce_layers), int(num_patches))
        tap_shapes = netG.__dict__.setdefault("_amx_tap_shapes", {})
        shapes = tap_shapes.get(skey)
        pre = {}
        capturing = reals.is_cuda and torch.cuda.is_current_stream_capturing()
        if shapes is not None and sample_ids is None and _PREDRAW and capturing:   # (eagerly the stream switches cost more than they return)
            side = _draw_stream(reals.device)

            def draw_all():
                main = torch.cuda.current_stream(reals.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    for l in sorted(sampled):
                        pre[l] = netF.draw_coords(sampled[l], shapes[l], num_patches, None, reals.device)
                        pre[l].record_stream(main)
            joined = []

            def sampler(i, shape):
                if not joined:
                    torch.cuda.current_stream(reals.device).wait_stream(side)
                    joined.append(1)
                if tuple(shape) != tuple(shapes[i]):
                    raise RuntimeError("contrastive step: the tap shapes changed under a cached sampling plan")
                return pre[i]
            if _LATE_DRAW:
                # a replayed graph hands its nodes to the device in capture order: six draw + filter launches in front of the forward
                # kept the first convolution waiting for ~170 us of launch latency; behind the first block they cost nothing
                sampler.on_start = draw_all
            else:
                draw_all()
        else:
            seen = {}

            def sampler(i, shape):
                seen[i] = tuple(shape)
                return netF.draw_coords(sampled[i], shape, num_patches, sample_ids, reals.device)
        out, rows, coords, dims = _train.forward_train_sampled(netG, reals, list(nce_layers), sampler)
        if shapes is None and sample_ids is None:
            tap_shapes[skey] = dict(seen)
        feat_sizes = dims
        feat_kq = None
    else:
        out, feat_kq = netG(reals, list(nce_layers), False)
        feat_sizes = [tuple(f.size()[2:]) for f in feat_kq]
    # The per-layer chains (sampling -> head -> loss, and their adjoints) are independent of each other and made of small
    # kernels that each occupy a fraction of the GPU: on CUDA they run on one stream per layer so they overlap -- autograd runs
    # a node's backward on the stream of its forward, so the adjoint chains overlap too -- and join before the sum.
    # (only while a HIP graph is being captured: launched eagerly, the extra stream switches cost the host more than the overlap
    # returns -- 16.9 vs 15.8 ms -- while a replayed graph gets the parallel branches for free: 12.6 -> 11.8 ms)
    # Inside a HIP graph the six chains run as ONE chain of batched launches (amx_mlp_heads_*, amx_supcon_loss_batch: every launch
    # serves all layers): a replayed graph pays ~8 us per dependent node whatever its size, and six chains of ~30 small nodes on
    # three streams were 1.5 ms of a 7 ms step.
    stacked = None
    capturing_now = reals.is_cuda and torch.cuda.is_current_stream_capturing()
    if sampled is not None and _BATCHED_HEADS and capturing_now:
        from . import supcon as _supcon
        pooled, ids = netF.forward_rows(rows, coords, None, batched=True)
        stacked = _supcon.batched_losses(criterions, pooled, seg_A, ids, feat_sizes)
    if stacked is not None:
        means = None
        layer_losses = list(stacked.detach().unbind(0))
        streams = None
    else:
        streams = _layer_streams(reals.device, len(feat_sizes)) if (reals.is_cuda and _PARALLEL_HEADS and capturing_now) else None
        ambient = torch.cuda.current_stream(reals.device) if streams is not None else None
        if sampled is not None:
            pooled, ids = netF.forward_rows(rows, coords, streams)
        else:
            pooled, ids = netF(feat_kq, num_patches, sample_ids, None, False, **({"streams": streams} if streams is not None else {}))
        means, layer_losses = [], []
        for k, (f_kq, sid, crit, layer, fsize) in enumerate(zip(pooled, ids, criterions, nce_layers, feat_sizes)):
            with (torch.cuda.stream(streams[k]) if streams is not None else contextlib.nullcontext()):
                m = crit(f_kq, seg_A, sid, torch.Size(fsize))
                if m.dim() != 0:                                  # (the HIP criterion returns the scalar: a mean of it would be three more launches)
                    m = m.mean()
            means.append(m)                                       # (kept alive until after the backward: no cross-stream reuse)
            layer_losses.append(m.detach())                       # the recorded per-layer loss IS this mean (it was reduced a second time)
        if streams is not None:
            for s in dict.fromkeys(streams):                      # (each distinct stream once: layers share streams)
                ambient.wait_stream(s)
    # total = sum_k mean_k * w_k * lambda_nce (supcl_model.py:815-843) as ONE weighted sum of the stacked means: the chain of scalar
    # multiplies and adds was a dozen 2-us launches on the main stream, forward and backward
    wkey = (str(reals.device), tuple(float(w) for w in nce_weights), float(lambda_nce), float(grad_accum_iters))
    wv = _WEIGHTS.get(wkey)
    if wv is None:                                            # (first built in an eager warm-up step: a host copy cannot be captured)
        wv = _WEIGHTS[wkey] = torch.tensor([w * lambda_nce / grad_accum_iters for w in nce_weights][: len(layer_losses)], dtype=torch.float32,
                                           device=reals.device)
    loss = ((stacked if stacked is not None else torch.stack(means)) * wv).sum()
    total = loss * grad_accum_iters if grad_accum_iters != 1 else loss
    (scaler.scale(loss) if scaler is not None else loss).backward()      # supcl_model.py:624-626
    return total, layer_losses, ids, out


def _sampled_route(netG, netF, reals, nce_layers, num_patches):
    """{module id: netF's layer index} when the step can take the sampled-tap route, else None: our Unet on its HIP training path
    with every tap at a conv id, our PatchSampleF, ascending distinct layer ids (netF's feature order = the forward's order)."""
    from ..model import train as _train
    from ..model.network import Unet
    from .patch_sample import PatchSampleF
    layers = [int(l) for l in nce_layers]
    if not (_SAMPLED_TAPS and isinstance(netG, Unet) and type(netF) is PatchSampleF and reals.is_cuda and num_patches > 0 and
            torch.is_grad_enabled() and layers == sorted(set(layers)) and not getattr(netG, "allow_torch_path", False)):
        return None
    # the route calls the training Function directly: only where Unet.forward would have routed the call there itself (network.py
    # forward: batch statistics or autograd through an instance-norm network) and nobody hooked the module's __call__
    if netG._forward_hooks or netG._forward_pre_hooks:
        return None
    wants_grad = reals.requires_grad or any(p.requires_grad for p in netG.parameters())
    norm = netG._cfg["norm"]
    if not ((norm == "batch" and (netG.training or wants_grad)) or (norm in ("instance", "instance_affine") and wants_grad)):
        return None
    try:
        if _train.sampled_unsupported_reason(netG, reals, layers) is not None:
            return None
    except Exception:
        return None
    return {l: k for k, l in enumerate(layers)}


def _total_norm(net):
    grads = [p.grad for p in net.parameters() if p.grad is not None]
    if not grads:
        return torch.zeros((), device=next(net.parameters()).device)
    return nn.utils.get_total_norm(grads, norm_type=2)


def _grad_norms(netG, netF):
    # clip_grad_norm_(max_norm=inf) only measures (supcl_model.py:635-655): the norm is what is recorded; its second half -- every
    # gradient multiplied by clamp(inf / norm, max=1) = 1 -- is an identity pass over all gradients and is left out
    if hasattr(nn.utils, "get_total_norm"):
        gG = [p.grad for p in netG.parameters() if p.grad is not None]
        gF = [p.grad for p in netF.parameters() if p.grad is not None]
        if gG and gF and hasattr(torch, "_foreach_norm") and all(g.is_cuda and g.dtype == torch.float32 and g.device == gG[0].device
                                                                   for g in gG + gF):
            # get_total_norm = vector_norm(stack(_foreach_norm(grads))) per network; the per-tensor norms of BOTH networks come from one
            # multi-tensor launch here (the same numbers, four launches fewer at the end of every step)
            norms = torch._foreach_norm(gG + gF, 2.0)
            return torch.linalg.vector_norm(torch.stack(norms[: len(gG)]), 2.0), torch.linalg.vector_norm(torch.stack(norms[len(gG):]), 2.0)
        return _total_norm(netG), _total_norm(netF)
    # older torch: the norm of the per-tensor norms (clip_grad_norm_(max_norm=inf) would also multiply every gradient by
    # clamp(inf / norm) -- NaN when the norm itself is inf)
    def total(net):
        norms = [torch.linalg.vector_norm(p.grad, 2) for p in net.parameters() if p.grad is not None]
        return torch.linalg.vector_norm(torch.stack(norms), 2) if norms else torch.zeros((), device=next(net.parameters()).device)
    return total(netG), total(netF)


def contrastive_step(netG, netF, criterions, real_A, real_B, seg_A, nce_layers, nce_weights=None, num_patches=512,
                     lambda_nce=1.0, optimizers=None, sample_ids=None, grad_accum_iters=1, grad_sync=None, do_step=None,
                     iters=None, grad_buckets=None, scaler=None):
    """Two aligned views through the shared network with feature taps, same-coordinate patch sampling, per-layer
    SupPatchNCELoss, weighted sum, backward and (optionally) the optimizer steps.

    netG: Unet (train mode: BatchNorm batch statistics over the two views, supcl_model.py:735-742);
    netF: PatchSampleF; criterions: one SupPatchNCELoss per nce layer; nce_weights default 1/len (supcl_model.py:388-393);
    optimizers: (opt_G, opt_F) or None (gradients only); sample_ids: captured coordinates per layer or None (randperm);
    grad_sync: callable run between backward and the optimizer steps (data parallel: the gradient all-reduce);
    grad_buckets: a ``data_parallel.GradientBuckets`` over (netG, netF) -- its ``sync()`` gathers the gradients into flat
    buckets and all-reduces them, its ``release()`` replaces ``optimizer.zero_grad()``.
    grad_accum_iters > 1 (supcl_model.py:618-661): the loss is divided by it on every call and the gradients accumulate; the
    optimizers step (and are zeroed) only on calls where ``do_step`` is true -- pass it directly, or pass the reference's
    running ``iters`` counter and it is ``iters % grad_accum_iters == 0``.  With optimizers and grad_accum_iters > 1 one of
    the two must be given: stepping on every call would shrink the gradients instead of accumulating them.
    scaler: a ``torch.amp.GradScaler`` -- the reference's loss-scaling protocol (supcl_model.py:523-525, 624-661): the loss is
    scaled before the backward; on a stepping call every optimizer is unscaled (so the recorded norms are those of the true
    gradients), ``scaler.step`` SKIPS an optimizer whose gradients hold an inf / NaN, and ``scaler.update()`` then lowers the
    scale.  With finite gradients the scale is a power of two and the step is bit-identical to the unscaled one.  (Eager steps
    only: GradScaler reads its found-inf flag on the host.)
    Returns an OrderedDict(loss, per_layer, grad_norm_G, grad_norm_F, sample_ids, out).
    """
    if scaler is not None and optimizers is None:
        raise ValueError("contrastive_step: scaler= needs the optimizers it unscales and steps")
    if grad_buckets is not None and grad_buckets.overlap and grad_accum_iters > 1:
        raise ValueError("GradientBuckets(overlap=True) reduces a bucket as soon as one backward filled it: not with grad_accum_iters > 1")
    if do_step is None:
        if iters is not None:
            do_step = iters % grad_accum_iters == 0
        elif optimizers is not None and grad_accum_iters > 1:
            raise ValueError("contrastive_step: with optimizers and grad_accum_iters > 1 pass do_step= or iters= "
                             "(the reference steps when iters % grad_accum_iters == 0, supcl_model.py:628)")
        else:
            do_step = True
    total, layer_losses, ids, out = _forward_backward(netG, netF, criterions, real_A, real_B, seg_A, nce_layers, nce_weights,
                                                     num_patches, lambda_nce, sample_ids, grad_accum_iters, scaler)
    if do_step:
        if grad_sync is not None:
            grad_sync()
        if grad_buckets is not None:
            grad_buckets.sync()
        if scaler is not None:
            for opt in optimizers:                   # unscale first: the norms below are those of the true gradients
                scaler.unscale_(opt)
    # the reference records the norms on stepping iterations only (supcl_model.py:631-655, after unscale_): on a non-stepping call the
    # gradients are partial sums still multiplied by the loss scale -- reported as NaN rather than as a number that means something else
    if do_step:
        gG, gF = _grad_norms(netG, netF)
    else:
        gG = gF = torch.full((), float("nan"), device=total.device)
    if optimizers is not None and do_step:
        for opt in optimizers:
            if scaler is not None:
                scaler.step(opt)                     # skipped when this optimizer's gradients hold an inf / NaN
            else:
                opt.step()
        if scaler is not None:
            scaler.update()
        if grad_buckets is not None:
            grad_buckets.release()
        else:
            for opt in optimizers:
                opt.zero_grad()
    elif grad_buckets is not None and do_step:
        grad_buckets.release()
    # ONE host synchronisation per step, after everything is enqueued (the reference reads every scalar with .item() as it
    # goes, supcl_model.py:841; that only paces the host, the values are the same)
    scalars = torch.stack([total.detach(), gG.detach(), gF.detach()] + layer_losses).tolist()
    per_layer = OrderedDict((str(layer), v) for layer, v in zip(nce_layers, scalars[3:]))
    # `out` detached: the backward has run, and a caller that keeps the record must not keep the autograd graph (and through it the
    # parameters' AccumulateGrad nodes, bound to this stream) alive -- a later HIP-graph capture of the same modules would then run
    # those nodes on the stream of THIS call, outside the capture
    return OrderedDict(loss=scalars[0], per_layer=per_layer, grad_norm_G=scalars[1], grad_norm_F=scalars[2], sample_ids=ids,
                       out=out.detach() if torch.is_tensor(out) else out)


class StepRecord(Mapping):
    """The record of one replayed step whose SCALARS (loss, per-layer losses, gradient norms) are read from the device on first use.
    ``GraphedContrastiveStep(lazy_scalars=True)`` enqueues a non-blocking copy of the step's scalar vector into pinned host memory and
    returns this mapping at once, so the host can enqueue the next step while this one still runs -- the reference reads every scalar
    with ``.item()`` as it goes (supcl_model.py:841), which idles the device between steps for as long as the host needs to come back
    (~0.2 ms of an 8 ms step here).  Reading ``rec["loss"]`` / ``["per_layer"]`` / ``["grad_norm_G"]`` / ``["grad_norm_F"]`` waits for
    that step only; ``["out"]`` and ``["sample_ids"]`` are device tensors of the step's static buffers (valid until the next replay)."""
    _KEYS = ("loss", "per_layer", "grad_norm_G", "grad_norm_F", "sample_ids", "out")

    def __init__(self, host, event, layers, sample_ids, out):
        self._host, self._event, self._layers, self._vals = host, event, layers, None
        self._fixed = {"sample_ids": sample_ids, "out": out}

    def resolve(self):
        if self._vals is None:
            self._event.synchronize()
            self._vals = self._host.tolist()                # the pinned slot is reused by a later step: keep the numbers, not the buffer
            self._host = None
        return self._vals

    def __getitem__(self, key):
        if key in self._fixed:
            return self._fixed[key]
        if key not in self._KEYS:
            raise KeyError(key)
        v = self.resolve()
        if key == "per_layer":
            return OrderedDict((str(layer), x) for layer, x in zip(self._layers, v[3:]))
        return v[{"loss": 0, "grad_norm_G": 1, "grad_norm_F": 2}[key]]

    def __iter__(self):
        return iter(self._KEYS)

    def __len__(self):
        return len(self._KEYS)


class GraphedContrastiveStep:
    """``contrastive_step`` captured once in a HIP graph and replayed: the step is ~700 kernel launches, most of them a few
    microseconds long; replaying them from one graph removes the launch gaps between them (16.3 -> 13.4 ms per step at 128^3
    on one MI355X, same box).  Static shapes only: every call must bring inputs of the shapes seen at capture.

    The first call runs ``warmup`` ordinary steps on a side stream (lazy module creation, allocator warm-up -- they are real
    training steps), then captures.  What goes into the graph: forward, sampling (drawn from torch's graph-safe generator, new
    coordinates every replay), heads, losses, backward and the gradient norms; the optimizer steps too when every optimizer was
    built with ``capturable=True`` and no ``grad_sync`` is given.  With ``grad_sync`` (data parallel: the gradient
    all-reduce) the graph ends after the backward; the all-reduce, the norms and the optimizers run eagerly after it.
    Returns the same OrderedDict as ``contrastive_step`` (ONE host synchronisation per call, for the scalars).

    Torch's rule for graph capture applies: nothing alive at capture time may still reference an autograd graph of these modules
    built on the default stream (e.g. a kept ``out`` / loss tensor of an eager step that has a ``grad_fn``) -- the parameters'
    AccumulateGrad nodes would run there, outside the capture.  ``contrastive_step`` returns detached records for that reason."""

    def __init__(self, netG, netF, criterions, nce_layers, optimizers, nce_weights=None, num_patches=512, lambda_nce=1.0,
                 grad_sync=None, warmup=3, grad_buckets=None, tail_graph=True, lazy_scalars=False):
        self.netG, self.netF, self.criterions, self.nce_layers = netG, netF, criterions, list(nce_layers)
        self.lazy_scalars, self._slots, self._slot_owner, self._slot_next = bool(lazy_scalars), None, None, 0
        self.optimizers, self.nce_weights, self.num_patches, self.lambda_nce = optimizers, nce_weights, num_patches, lambda_nce
        self.grad_buckets = grad_buckets
        if grad_buckets is not None and getattr(grad_buckets, "overlap", False):
            # the post-accumulate hooks of overlap mode would launch RCCL all-reduces and mutate Python bookkeeping DURING the capture;
            # neither re-runs on replay, so sync() would skip those buckets
            raise ValueError("GraphedContrastiveStep needs GradientBuckets(overlap=False): hooks that launch collectives cannot be captured")
        if grad_buckets is not None and grad_sync is None and grad_buckets.world > 1:
            grad_sync = grad_buckets.sync
        self.grad_sync, self.warmup = grad_sync, warmup
        self.graph = None
        self.tail_graph = None
        capturable = optimizers is None or all(o.defaults.get("capturable", False) for o in optimizers)
        self.opt_in_graph = grad_sync is None and optimizers is not None and capturable
        # data parallel: [graph: forward + backward] -> gradient all-reduce (eager, RCCL) -> [second graph: norms + optimizers]
        self.tail_in_own_graph = bool(tail_graph) and grad_sync is not None and capturable

    def _eager(self):
        return _forward_backward(self.netG, self.netF, self.criterions, self.A, self.B, self.seg, self.nce_layers,
                                 self.nce_weights, self.num_patches, self.lambda_nce, None, 1)

    def _tail(self, total, layer_losses):
        """gradient norms (+ optimizers): inside the graph when possible, else eagerly after the replay."""
        gG, gF = _grad_norms(self.netG, self.netF)
        if self.optimizers is not None:
            for opt in self.optimizers:
                opt.step()
        return torch.stack([total.detach(), gG.detach(), gF.detach()] + layer_losses)

    def _capture(self, real_A, real_B, seg_A):
        # autograd graphs of earlier eager steps that are only held by reference cycles still own the parameters' AccumulateGrad nodes,
        # bound to the stream of those steps (usually the default stream, which cannot join a capture): collect them first
        gc.collect()
        self.A, self.B, self.seg = real_A.clone(), real_B.clone(), seg_A.clone()
        if all(t.is_cuda for t in (real_A, real_B, seg_A)):
            # (loads the multi-tensor copy kernel now: its first use would otherwise be the first replayed step, ~30 ms late)
            torch._foreach_copy_([self.A, self.B, self.seg], [real_A, real_B, seg_A])
        side = torch.cuda.Stream(device=self.A.device)
        side.wait_stream(torch.cuda.current_stream(self.A.device))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self._zero()
                total, layer_losses, _, _ = self._eager()
                if self.grad_sync is not None:
                    self.grad_sync()
                self._tail(total, layer_losses)
        torch.cuda.current_stream(self.A.device).wait_stream(side)
        self._zero()
        # hyper-parameters at capture time: FusedAdamW mirrors are refreshed (and waited for) BEFORE the capture starts -- inside it a
        # synchronise would abort the capture; stock optimizers get the snapshot of what is being frozen into the graph
        for opt in (self.optimizers or ()):
            if hasattr(opt, "refresh_hyperparameters"):
                opt.refresh_hyperparameters()
            else:
                self._check_frozen_hyperparameters(opt)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.total, self.layer_losses, self.ids, out = self._eager()
            self.out = out.detach() if torch.is_tensor(out) else out
            if self.grad_buckets is not None:
                self.grad_buckets.collect()                 # part of every replay: fresh gradients -> the flat buckets
            if self.opt_in_graph or (self.grad_sync is None and self.optimizers is None):
                self.scalars = self._tail(self.total, self.layer_losses)
        if self.tail_in_own_graph:
            self.tail_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.tail_graph):
                self.tail_scalars = self._tail(self.total, self.layer_losses)

    def _check_frozen_hyperparameters(self, opt):
        """A stock capturable optimizer inside a graph replays the hyper-parameters of the capture: refuse to diverge silently."""
        snap = tuple((g["lr"] if not torch.is_tensor(g["lr"]) else None, tuple(g["betas"]), g["eps"], g["weight_decay"]) for g in opt.param_groups)
        seen = self.__dict__.setdefault("_hyper_snap", {})       # first call: from _capture, i.e. the values frozen into the graph
        if seen.setdefault(id(opt), snap) != snap:
            raise RuntimeError("GraphedContrastiveStep: the hyper-parameters of a captured torch optimizer changed (lr / betas / eps / "
                               "weight_decay are frozen into the graph); use anatomix_amd.pretraining.FusedAdamW, a tensor lr, or build a "
                               "new GraphedContrastiveStep")

    def _zero(self):
        if self.grad_buckets is not None:
            self.grad_buckets.release()
            return
        for net in (self.netG, self.netF):
            for p in net.parameters():
                p.grad = None

    def __call__(self, real_A, real_B, seg_A):
        first = self.graph is None
        if first:
            self._capture(real_A, real_B, seg_A)
        else:
            # one launch for the three static input buffers (three eager copies were three host-paced launches in front of every replay)
            srcs = [real_A, real_B, seg_A]
            dsts = [self.A, self.B, self.seg]
            if all(t.is_cuda and t.device == d.device and t.dtype == d.dtype and t.shape == d.shape for t, d in zip(srcs, dsts)):
                torch._foreach_copy_(dsts, srcs)
            else:
                for d, t in zip(dsts, srcs):
                    d.copy_(t)
        # optimizer.step() does not run on a replay: hand the CURRENT param_groups values (lr schedulers change them every epoch,
        # base_model.py update_learning_rate) to the captured step through the optimizers' host mirrors
        for opt in self.optimizers or ():
            if hasattr(opt, "refresh_hyperparameters"):
                opt.refresh_hyperparameters()
            elif self.opt_in_graph or self.tail_graph is not None:
                self._check_frozen_hyperparameters(opt)
        self.graph.replay()
        if hasattr(self, "scalars") and (self.opt_in_graph or self.optimizers is None):
            scalars = self.scalars
        else:
            if self.grad_sync is not None:
                self.grad_sync()
            if self.tail_graph is not None:
                self.tail_graph.replay()
                scalars = self.tail_scalars
            else:
                scalars = self._tail(self.total, self.layer_losses)
        if self.lazy_scalars and scalars.is_cuda:
            # a ring of pinned slots; a slot about to be reused first hands its numbers to the record that still owns it
            if self._slots is None:
                self._slots = [torch.empty(scalars.shape, dtype=scalars.dtype).pin_memory() for _ in range(8)]
                self._slot_owner = [None] * len(self._slots)
            k = self._slot_next
            self._slot_next = (k + 1) % len(self._slots)
            owner = self._slot_owner[k]() if self._slot_owner[k] is not None else None
            if owner is not None:
                owner.resolve()
            self._slots[k].copy_(scalars, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(scalars.device))
            rec = StepRecord(self._slots[k], ev, self.nce_layers, self.ids, self.out)
            self._slot_owner[k] = weakref.ref(rec)
            return rec
        vals = scalars.tolist()
        per_layer = OrderedDict((str(layer), v) for layer, v in zip(self.nce_layers, vals[3:]))
        return OrderedDict(loss=vals[0], per_layer=per_layer, grad_norm_G=vals[1], grad_norm_F=vals[2], sample_ids=self.ids,
                           out=self.out)
