"""Feature-extraction front end of the registration pipeline with the surface of
``anatomix.registration.convex_adam_utils`` (reference lines 16-78, 134-221): model loading,
min-max normalisation and the sliding-window feature extraction of the fixed / moving volumes.
The rest of that reference module (MIND-SSC, correlation volume, coupled-convex solver, Jacobian
utilities) is downstream of the UNet path and out of scope here.
"""
from __future__ import annotations

import numpy as np
import torch

from ..model.load_from_hf import ANATOMIX_VARIANTS, _load_handling_compile
from ..model.network import Unet
from .sliding_window import sliding_window_inference


def load_model(ckpt_path=None, hf_variant=None, output_nc=16, num_downs=4, ngf=16, norm="batch", interp="nearest",
               pooling="Max", device=None):
    """convex_adam_utils.py:16-78: build the Unet (positional argument order of the reference), load a
    checkpoint (plain state_dict, optionally ``_orig_mod.``-prefixed) and put it in eval mode on `device`."""
    if hf_variant is not None:
        kw = ANATOMIX_VARIANTS[hf_variant]["unet_kwargs"]
        model = Unet(**kw)
    else:
        model = Unet(3, 1, output_nc, num_downs, ngf=ngf, norm=norm, interp=interp, pooling=pooling)
    if ckpt_path is not None:
        _load_handling_compile(model, torch.load(ckpt_path, map_location="cpu"))
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
