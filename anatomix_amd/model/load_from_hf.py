"""Variant registry and checkpoint loading with the surface of
``anatomix.model.load_from_hf`` (reference: anatomix/model/load_from_hf.py:11-79).

The Hub download itself needs network access; everything after it (variant -> constructor kwargs,
``_orig_mod.`` / ``module.`` prefix stripping, ``load_state_dict(strict=True)``) is reproduced so that a local
``.pth`` can be loaded with ``load_from_hf(variant, weights_path=...)``.
"""
from __future__ import annotations

import warnings

import torch

from .network import Unet

DEFAULT_REPO = "neeldey/anatomix"

# load_from_hf.py:11-36 -- constructor kwargs per published variant.
ANATOMIX_VARIANTS = {
    "anatomix": {
        "unet_kwargs": dict(dimension=3, input_nc=1, output_nc=16, num_downs=4, ngf=16),
        "output_channels": 16,
    },
    "anatomix-dev": {
        "unet_kwargs": dict(dimension=3, input_nc=1, output_nc=32, num_downs=5, ngf=32, norm="instance",
                            pooling="Avg", interp="trilinear", norm_eps=1e-2),
        "output_channels": 32,
    },
    # load_from_hf.py:25-35 -- the 3D ViT (PrimusV2-S); body restated, see anatomix_amd/model/vit3d/architectures.py
    "anatomix-dev-vit": {
        "vit_kwargs": dict(input_channels=1, num_classes=32, embed_dim=396, eva_depth=12, eva_numheads=6,
                           patch_embed_size=(8, 8, 8), input_shape=(128, 128, 128), num_register_tokens=8, init_values=0.1,
                           scale_attn_inner=True, qk_norm=True, out_norm="demean", out_norm_eps=1e-2, register_init_std=0.02,
                           in_eps=1e-2),
        "output_channels": 32,
    },
}


def convert_dict(state_dict):
    """pretraining/models/base_model.py:458-466: strips the ``module.`` (nn.DataParallel) and ``_orig_mod.`` (torch.compile) parts
    of every key -- wherever they sit, e.g. ``_orig_mod.module.model.0.weight`` -- so that the trainer's ``*_net_G.pth`` files load
    into a bare network whatever wrapper they were saved through.  Plain state dicts pass through untouched."""
    out = type(state_dict)() if isinstance(state_dict, dict) else {}
    for k, v in state_dict.items():
        out[k.replace("module.", "").replace("_orig_mod.", "")] = v
    return out


def _load_handling_compile(model, state_dict):
    """load_from_hf.py:39-49 accepts checkpoints saved from a torch.compile()-wrapped module (``_orig_mod.`` prefix); the trainer's
    own loader (base_model.py:340-349) also accepts nn.DataParallel's ``module.`` prefix, decided -- as there -- on the FIRST key.
    Both are handled here, so the published ``<variant>.pth`` files and the pretraining checkpoints load through one function."""
    if state_dict:
        first = next(iter(state_dict))
        if "module." in first or "_orig_mod." in first:
            state_dict = convert_dict(state_dict)
    if hasattr(state_dict, "_metadata"):              # base_model.py:347-348
        del state_dict._metadata
    model.load_state_dict(state_dict, strict=True)
    return model


def build_variant(variant):
    """Un-initialised model of a registered variant."""
    if variant not in ANATOMIX_VARIANTS:
        raise ValueError(f"Unknown variant {variant!r}. Known: {sorted(ANATOMIX_VARIANTS)}")
    config = ANATOMIX_VARIANTS[variant]
    if "vit_kwargs" in config:                          # load_from_hf.py:74-76
        from .vit3d import PrimusV2
        return PrimusV2(**config["vit_kwargs"])
    return Unet(**config["unet_kwargs"])


def load_from_hf(variant, repo_id=DEFAULT_REPO, revision=None, map_location="cpu", weights_path=None):
    """load_from_hf.py:52-79.  ``weights_path`` (extension) loads a local ``<variant>.pth`` instead of
    downloading it."""
    model = build_variant(variant)
    if "vit_kwargs" in ANATOMIX_VARIANTS[variant]:
        # The reference builds this variant on `dynamic-network-architectures==0.4.4` + timm, neither of which was available to
        # compare against: module names / state_dict keys, token order and the rotary table of anatomix_amd.model.vit3d restate the
        # published architecture and are NOT pinned against the upstream package.
        warnings.warn(f"{variant}: the PrimusV2 body of anatomix_amd is an UNPINNED restatement of the upstream package "
                      "(dynamic-network-architectures 0.4.4 / timm were unavailable); a published checkpoint may not load "
                      "(strict key match) and, if it loads, its features are not verified against the reference", stacklevel=2)
    if weights_path is None:
        from huggingface_hub import hf_hub_download   # needs network access
        weights_path = hf_hub_download(repo_id, f"{variant}.pth", revision=revision)
    state_dict = torch.load(weights_path, map_location=map_location)
    return _load_handling_compile(model, state_dict)
