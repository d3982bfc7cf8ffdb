"""The optimizer of the contrastive step: ``torch.optim.AdamW`` as the reference builds it for netG and for netF
(pretraining/models/supcl_model.py:510-516, 584-590; stepped at :628-661), with every parameter tensor of an optimizer updated
by ONE HIP launch (``amx_adamw_step``) instead of torch's few kernels per parameter -- in the graph-replayed step the stock
capturable AdamW is ~170 launches of a few microseconds each.

Same constructor arguments, same update rule, same ``state_dict`` layout as ``torch.optim.AdamW(capturable=True)`` (per parameter:
``step`` a 0-dim fp32 device tensor, ``exp_avg``, ``exp_avg_sq``), so a checkpoint moves between the two.  GPU only: fp32
parameters on a CUDA device -- anything else raises, there is no host path."""
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
        for group in self.param_groups:
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
            beta1, beta2 = group["betas"]
            for dev, (rows, steps, keep) in per_device.items():
                with torch.cuda.device(dev):
                    torch._foreach_add_(steps, 1.0)                                   # t: one launch for the whole list
                    table = np.asarray(rows, dtype=np.int64)
                    _lib.check(lib.amx_adamw_step(table.ctypes.data_as(ctypes.c_void_p), len(rows), float(group["lr"]), float(beta1),
                                                  float(beta2), float(group["eps"]), float(group["weight_decay"]),
                                                  int(bool(group["maximize"])),
                                                  ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return loss
