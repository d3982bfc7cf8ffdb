"""3D ViT (Primus; arXiv:2503.01835) backbones with the surface of ``anatomix.model.vit3d``."""
from .architectures import PRIMUS_CONFIGS, PrimusV2, ChannelDemean, ChannelLayerNorm, build_out_norm

__all__ = ["PrimusV2", "PRIMUS_CONFIGS", "ChannelDemean", "ChannelLayerNorm", "build_out_norm"]
