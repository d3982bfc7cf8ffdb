from .network import Unet, get_norm_layer, get_actvn_layer  # noqa: F401
