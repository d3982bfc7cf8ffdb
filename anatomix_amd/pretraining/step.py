"""One contrastive-pretraining step: the per-batch body of the reference's SupCLModel
(forward supcl_model.py:723-770, calculate_NCE_loss 801-843, optimize_parameters 603-661)."""
from collections import OrderedDict

import torch
import torch.nn as nn


def contrastive_step(netG, netF, criterions, real_A, real_B, seg_A, nce_layers, nce_weights=None, num_patches=512,
                     lambda_nce=1.0, optimizers=None, sample_ids=None, grad_accum_iters=1, grad_sync=None):
    """Two aligned views through the shared network with feature taps, same-coordinate patch sampling, per-layer
    SupPatchNCELoss, weighted sum, backward and (optionally) the optimizer steps.

    netG: Unet (train mode: BatchNorm batch statistics over the two views, supcl_model.py:735-742);
    netF: PatchSampleF; criterions: one SupPatchNCELoss per nce layer; nce_weights default 1/len (supcl_model.py:388-393);
    optimizers: (opt_G, opt_F) or None (gradients only); sample_ids: captured coordinates per layer or None (randperm);
    grad_sync: callable run between backward and the optimizer steps (data parallel: the gradient all-reduce).
    Returns an OrderedDict(loss, per_layer, grad_norm_G, grad_norm_F, sample_ids, out).
    """
    if nce_weights is None:
        nce_weights = [1.0 / len(nce_layers)] * len(nce_layers)
    reals = torch.cat((real_A, real_B), dim=0) if real_B is not None else real_A
    out, feat_kq = netG(reals, list(nce_layers), False)
    pooled, ids = netF(feat_kq, num_patches, sample_ids, None, False)
    total = 0.0
    layer_losses = []
    for f_kq, sid, crit, layer, w, feat in zip(pooled, ids, criterions, nce_layers, nce_weights, feat_kq):
        loss = crit(f_kq, seg_A, sid, feat.size()[2:])
        total = total + loss.mean() * w * lambda_nce
        layer_losses.append(loss.detach().mean())
    (total / grad_accum_iters).backward()
    if grad_sync is not None:
        grad_sync()
    # clip_grad_norm_(max_norm=inf) only measures (supcl_model.py:635-655)
    gG = nn.utils.clip_grad_norm_(netG.parameters(), max_norm=float("inf"), norm_type=2)
    gF = nn.utils.clip_grad_norm_(netF.parameters(), max_norm=float("inf"), norm_type=2)
    if optimizers is not None:
        for opt in optimizers:
            opt.step()
        for opt in optimizers:
            opt.zero_grad()
    # ONE host synchronisation per step, after everything is enqueued (the reference reads every scalar with .item() as it
    # goes, supcl_model.py:841; that only paces the host, the values are the same)
    scalars = torch.stack([total.detach(), gG.detach(), gF.detach()] + layer_losses).tolist()
    per_layer = OrderedDict((str(layer), v) for layer, v in zip(nce_layers, scalars[3:]))
    return OrderedDict(loss=scalars[0], per_layer=per_layer, grad_norm_G=scalars[1], grad_norm_F=scalars[2], sample_ids=ids,
                       out=out)
