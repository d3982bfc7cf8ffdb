"""anatomix_amd -- MI355X (gfx950) native 3D-UNet feature-extraction path with the
``anatomix.model.network.Unet`` surface of neel-dey/anatomix."""
from .model.network import Unet, get_norm_layer, get_actvn_layer  # noqa: F401
from .model.load_from_hf import ANATOMIX_VARIANTS, load_from_hf, build_variant  # noqa: F401

__version__ = "0.1.0"
