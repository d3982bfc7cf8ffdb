"""The optimizer of the contrastive step: ``torch.optim.AdamW`` as the reference builds it for netG and for netF
(pretraining/models/supcl_model.py:510-516, 584-590; stepped at :628-661), with every parameter tensor of an optimizer updated
by ONE HIP launch (``amx_adamw_step``) instead of torch's few kernels per parameter -- in the graph-replayed step the stock
capturable AdamW is ~170 launches of a few microseconds each.

Same constructor arguments, same update rule, same ``state_dict`` layout as ``torch.optim.AdamW(capturable=True)`` (per parameter:
``step`` a 0-dim fp32 device tensor, ``exp_avg``, ``exp_avg_sq``), so a checkpoint moves between the two.  GPU only: fp32
parameters on a CUDA device -- anything else raises, there is no host path.

Hyper-parameters and HIP graphs.  The reference changes ``lr`` every epoch (base_model.py: get_scheduler / update_learning_rate
set ``param_group['lr']``).  Kernel arguments passed by value are frozen into a captured graph, so the kernel reads {lr, betas, eps,
weight_decay} from a small per-group DEVICE buffer, which ``step()`` refreshes from a pinned host mirror with a copy that is part
of the stream (and of a capture): ``GraphedContrastiveStep`` calls ``refresh_hyperparameters()`` -- a host write into that mirror --
before every replay, and the replayed copy carries the current ``param_groups`` values to the kernel."""
import ctypes

import numpy as np
import torch

from .. import _lib


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *, maximize=False,
                 foreach=None, capturable=True, differentiable=False, fused=None):
        if isinstance(lr, torch.Tensor):
            raise ValueError("FusedAdamW: lr must be a Python number (schedulers set group['lr'] to one)")
        if amsgrad:
            raise NotImplementedError("FusedAdamW: amsgrad is not offered (the reference does not use it, supcl_model.py:510-516)")
        if differentiable:
            raise NotImplementedError("FusedAdamW: differentiable=True is not offered")
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        # `capturable` is what GraphedContrastiveStep looks at: the step count lives on the device, a step is graph-safe
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=maximize, foreach=None,
                        capturable=True, differentiable=False, fused=None, decoupled_weight_decay=True)
        super().__init__(params, defaults)
        self._hyper = {}            # (group index, device) -> (pinned host mirror [5] float64, device buffer [5] float64)

    @staticmethod
    def _hyper_values(group):
        lr = group["lr"]
        if isinstance(lr, torch.Tensor):
            raise ValueError("FusedAdamW: lr must be a Python number (schedulers set group['lr'] to one)")
        beta1, beta2 = group["betas"]
        vals = (float(lr), float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]))
        if not (vals[0] >= 0.0 and vals[3] >= 0.0 and 0.0 <= vals[1] < 1.0 and 0.0 <= vals[2] < 1.0 and vals[4] >= 0.0):
            raise ValueError(f"FusedAdamW: hyper-parameters out of range: lr {vals[0]}, betas ({vals[1]}, {vals[2]}), eps {vals[3]}, "
                             f"weight_decay {vals[4]}")
        return vals

    def refresh_hyperparameters(self):
        """Write the current ``param_groups`` values into the pinned host mirrors (host-side only; nothing is enqueued).  Called by
        ``step()``, and -- because ``step()`` does not run when a captured graph is replayed -- by ``GraphedContrastiveStep`` before
        every replay.  A CHANGED value first waits for the device: an earlier replay may not have read the mirror yet."""
        for gi, group in enumerate(self.param_groups):
            vals = self._hyper_values(group)
            for (g2, dev), (host, _) in self._hyper.items():
                if g2 == gi and tuple(host.tolist()) != vals:
                    # (not while the stream is being captured: a device-wide synchronise is illegal there and nothing can be
                    #  replaying -- hence reading the mirror -- before the capture ends)
                    if not torch.cuda.is_current_stream_capturing():
                        torch.cuda.synchronize(dev)
                    host.copy_(torch.tensor(vals, dtype=torch.float64))

    def _hyper_buffers(self, gi, group, dev):
        key = (gi, dev)
        if key not in self._hyper:
            host = torch.tensor(self._hyper_values(group), dtype=torch.float64).pin_memory()
            self._hyper[key] = (host, torch.empty(5, dtype=torch.float64, device=dev))
        return self._hyper[key]

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        elif not (torch.is_tensor(st["step"]) and st["step"].is_cuda):
            # a state_dict written by a non-capturable torch.optim.AdamW keeps its step count on the host
            st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32).to(p.device)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        self.refresh_hyperparameters()
        for gi, group in enumerate(self.param_groups):
            per_device = {}
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("FusedAdamW does not support sparse gradients")
                if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError(f"FusedAdamW: contiguous fp32 parameters on a CUDA device only (got {p.dtype} on {p.device}, "
                                       f"gradient {g.dtype})")
                st = self._state_of(p)
                if not g.is_contiguous():
                    g = g.contiguous()
                rows, steps, keep = per_device.setdefault(p.device, ([], [], []))
                rows.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(),
                             p.numel()))
                steps.append(st["step"])
                keep.append(g)
            for dev, (rows, steps, keep) in per_device.items():
                with torch.cuda.device(dev):
                    host, dbuf = self._hyper_buffers(gi, group, dev)
                    dbuf.copy_(host, non_blocking=True)                               # captured with the step: replays re-read the mirror
                    torch._foreach_add_(steps, 1.0)                                   # t: one launch for the whole list
                    table = np.asarray(rows, dtype=np.int64)
                    _lib.check(lib.amx_adamw_step_dev(table.ctypes.data_as(ctypes.c_void_p), len(rows), _lib.ptr(dbuf),
                                                      int(bool(group["maximize"])),
                                                      ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return loss
