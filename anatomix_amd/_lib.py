"""ctypes binding of the C ABI in ``include/anatomix_amd.h`` (libanatomix_amd.so, built in-tree by
``anatomix_amd/csrc/Makefile``).  There is no CPU implementation behind these symbols: if the
shared library is missing the import of the HIP path fails loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AMX_LIB_PATH (read here, by the Python loader, never by the library): another build of the same sources, e.g. the
# -DAMX_EXPERIMENT library of `make exp` whose ablation switches tools/ drives through the environment
LIB_PATH = os.environ.get("AMX_LIB_PATH") or os.path.join(_HERE, "csrc", "libanatomix_amd.so")


def exp_env(name, default):
    """A/B switches of the Python side (tools/ only): read from the environment ONLY under AMX_EXPERIMENT=1, like the C++ sources'
    exp_env() behind -DAMX_EXPERIMENT; in a product run every switch is its default whatever the environment holds."""
    if os.environ.get("AMX_EXPERIMENT", "0") != "1":
        return default
    return os.environ.get(name, default)

NORM = {"none": 0, "batch": 1, "instance": 2, "instance_affine": 3}
ACT = {"none": 0, "relu": 1, "lrelu": 2}
POOL = {"Max": 0, "Avg": 1}
INTERP = {"nearest": 0, "trilinear": 1}
# "strict" = bf16x2: split hi + lo operands, fp32-grade results with fp32's exponent range (include/anatomix_amd.h)
# "f16x2mx" (alias "strict_mx") = f16 hi + lo pairs whose correction products run on the block-scaled fp8 MFMA (2 instead of 3 MFMA-
# equivalents per product); implemented for the InstanceNorm configurations, where it is the default (Unet.precision)
PRECISION = {"f16": 0, "fp16": 0, "float16": 0, "bf16": 1, "bfloat16": 1, "f16x2": 2, "bf16x2": 3, "strict": 3, "fp32": 3,
             "f16x2mx": 4, "strict_mx": 4}


class UnetCfg(C.Structure):
    _fields_ = [
        ("input_nc", C.c_int32), ("output_nc", C.c_int32), ("num_downs", C.c_int32), ("ngf", C.c_int32),
        ("norm", C.c_int32), ("norm_eps", C.c_float), ("activation", C.c_int32), ("act_slope", C.c_float),
        ("final_act", C.c_int32), ("pooling", C.c_int32), ("interp", C.c_int32), ("doubleconv", C.c_int32),
        ("use_skip", C.c_int32), ("precision", C.c_int32),
    ]


class VitCfg(C.Structure):
    _fields_ = [
        ("input_channels", C.c_int32), ("num_classes", C.c_int32), ("embed_dim", C.c_int32), ("depth", C.c_int32),
        ("heads", C.c_int32), ("num_register_tokens", C.c_int32), ("grid_d", C.c_int32), ("grid_h", C.c_int32),
        ("grid_w", C.c_int32), ("hidden", C.c_int32), ("dec1", C.c_int32), ("dec2", C.c_int32), ("qk_norm", C.c_int32),
        ("scale_attn_inner", C.c_int32), ("layer_scale", C.c_int32), ("in_eps", C.c_float), ("out_norm", C.c_int32),
        ("decoder_split", C.c_int32), ("stem_split", C.c_int32),
    ]


_P = C.c_void_p
_I = C.c_int


class PackReq(C.Structure):
    _fields_ = [("d_weight", C.c_void_p), ("d_wpk", C.c_void_p), ("weight_mode", C.c_int32), ("cin_real", C.c_int32),
                ("cin_pad", C.c_int32), ("cout_real", C.c_int32), ("cout", C.c_int32), ("w", C.c_int32)]


WEIGHTS_PREPACKED = 16


class LaunchRecord(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("module_idx", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("n", C.c_int32), ("d", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("ms", C.c_float),
                ("flops", C.c_double), ("bytes", C.c_double)]


_P = C.c_void_p
_I = C.c_int
SYMBOLS = {
    "amx_version": (_I, []),
    "amx_debug_fill_lds": (_I, [C.c_uint, _P]),
    "amx_last_error": (C.c_char_p, []),
    "amx_unet_create": (_I, [C.POINTER(_P), C.POINTER(UnetCfg)]),
    "amx_unet_destroy": (None, [_P]),
    "amx_unet_num_modules": (_I, [_P]),
    "amx_unet_num_convs": (_I, [_P]),
    "amx_unet_conv_info": (_I, [_P, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "amx_unet_load_conv": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "amx_unet_workspace_bytes": (C.c_size_t, [_P, _I, _I, _I, _I]),
    "amx_unet_forward": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, C.c_size_t, _P]),
    "amx_unet_forward_profiled": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, C.c_size_t, _P, C.POINTER(LaunchRecord), _I,
                                        C.POINTER(_I)]),
    "amx_unet_module_info": (_I, [_P, _I, C.POINTER(_I), C.POINTER(_I)]),
    "amx_unet_numerics_status": (_I, [_P, _I, _P]),
    "amx_unet_forward_taps": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, C.c_size_t, C.POINTER(_I), _I, C.POINTER(_P), _I, _P]),
    "amx_unet_forward_window": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, C.c_size_t, _P]),
    "amx_unet_forward_windows": (_I, [_P, _P, _I, _I, _I, _I, C.POINTER(C.c_int), _I, _I, _I, _P, _P, _P, C.c_size_t, _P]),
    "amx_unet_forward_windows_pipelined": (_I, [_P, _P, _I, _I, _I, _I, C.POINTER(C.c_int), _I, _I, _I, _P, _P, _P, C.c_size_t, _I, _P]),
    "amx_sw_normalize": (_I, [_P, _P, _I, C.c_longlong, _P]),
    "amx_sw_count": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "amx_conv3d_packed_bytes": (C.c_size_t, [_I, _I]),
    "amx_conv3d_k3_reflect": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, C.c_float, _I, _P, _P, _P, _P]),
    "amx_conv3d_dgrad_interior_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "amx_conv3d_dgrad_interior": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "amx_conv3d_dgrad_shell_scratch_bytes": (C.c_size_t, []),
    "amx_conv3d_dgrad_fold_shell": (_I, [_P, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "amx_conv3d_pack_batch": (_I, [C.POINTER(PackReq), _I, _I, _P]),
    "amx_conv3d_scratch_bytes": (C.c_size_t, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "amx_conv3d_k3_reflect_ws": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, C.c_float, _I, _P, _P, _P, _P, C.c_size_t, _P]),
    "amx_conv3d_upcat_merged_packed_bytes": (C.c_size_t, [_I, _I, _I]),
    "amx_conv3d_upcat_merged": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, C.c_float, _I, _P, _P, _P, _P]),
    "amx_conv3d_k3_reflect_ex": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, C.c_float, _I, _P, _P, _P, _P]),
    "amx_pool2": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "amx_instance_norm_scratch_bytes": (C.c_size_t, [_I, _I]),
    "amx_instance_norm": (_I, [_P, _P, _P, C.c_float, _I, C.c_longlong, _I, _I, C.c_float, _P, _I, _P]),
    "amx_upsample2_trilinear": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "amx_upsample2_trilinear_backward": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "amx_upcat_split_backward": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "amx_upcat_split_backward_framed": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "amx_export_ncdhw": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "amx_import_ncdhw": (_I, [_P, _P, _I, _I, _I, _I, _I, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, _I, _I, _P]),
    "amx_train_scratch_bytes": (C.c_size_t, [_I]),
    "amx_bn_train_forward": (_I, [_P, _P, _P, _P, C.c_float, _I, C.c_longlong, _I, _I, C.c_float, _P, _P, _P, _P, _P,
                                  C.c_float, _I, _P]),
    "amx_bn_act_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, C.c_float, _P, _I, _P]),
    "amx_bn_act_backward_recompute": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, C.c_float, _P, _I, _P]),
    "amx_pad_fold": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "amx_adamw_step": (_I, [_P, _I, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _I, _P]),
    "amx_adamw_step_dev": (_I, [_P, _I, _P, _I, _P]),
    "amx_pool2_max_backward": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "amx_conv3d_wgrad_scratch_bytes": (C.c_size_t, [_I, _I, _I, _I, _I, _I]),
    "amx_conv3d_wgrad": (_I, [_P, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, _P, _I, _P, _I, _I, _I, _I, _I, _I,
                              _I, _P, _I, _P, C.c_size_t, _I, _P]),
    "amx_attention_scratch_bytes": (C.c_size_t, [_I, _I, _I, _I]),
    "amx_attention_qknorm_rope": (_I, [_P, _P, _P, _P, _P, _P, _P, C.c_float, _P, _I, _I, _I, _I, _I, _P, _P, C.c_size_t, _P]),
    "amx_attention_prepared": (_I, [_P, C.c_size_t, _I, _I, _I, _I, _P, _P]),
    "amx_vit_create": (_I, [C.POINTER(_P), C.POINTER(VitCfg)]),
    "amx_vit_destroy": (None, [_P]),
    "amx_vit_num_params": (_I, [_P]),
    "amx_vit_param_name": (C.c_char_p, [_P, _I]),
    "amx_vit_load": (_I, [_P, C.POINTER(_P), _I, _P, _P]),
    "amx_vit_workspace_bytes": (C.c_size_t, [_P, _I]),
    "amx_vit_forward": (_I, [_P, _P, _P, _I, _P, C.c_size_t, _I, _P]),
    "amx_vit_debug_read": (_I, [_P, C.c_char_p, _P, C.c_size_t, C.POINTER(C.c_size_t), _P]),
    "amx_linear": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "amx_supcon_scratch_bytes": (C.c_size_t, [_I, _I]),
    "amx_supcon_loss": (_I, [_P, _P, _I, _I, C.c_float, _I, _I, _I, _P, _P, _P, C.c_size_t, _P]),
    "amx_mlp_head_forward": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, C.c_float, C.c_float, _I, C.c_float, _P, _P, _P, _P, _P]),
    "amx_mlp_head_scratch_bytes": (C.c_size_t, [_I, _I, _I]),
    "amx_mlp_head_backward": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _I, C.c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "amx_mlp_heads_forward": (_I, [_I, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, C.c_float, C.c_float, _I, C.c_float, _P, _P, _P, _P, _P]),
    "amx_mlp_heads_backward": (_I, [_I, _P, _P, _I, _P, _I, _I, _P, _P, _I, C.c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "amx_supcon_loss_batch": (_I, [_I, _P, _P, _I, _I, C.c_float, _I, _I, _I, _P, _P, _P, C.c_size_t, _P]),
    "amx_gather_labels_batch": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P]),
    "amx_conv3d_backward_sampled_scratch_bytes": (C.c_size_t, [_I]),
    "amx_conv3d_backward_sampled": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, C.c_size_t, _I, _P]),
    "amx_gather_rows": (_I, [_P, _I, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, _P, _I, _I, _I, _P, _P]),
    "amx_scatter_rows": (_I, [_P, _P, _P, _I, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, _I, _I, _I, _I, _P]),
    "amx_sample_coords": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "amx_sample_perm": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "amx_import_input": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "amx_gather_labels": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P]),
    "amx_mindssc_scratch_bytes": (C.c_size_t, [_I, _I, _I]),
    "amx_mindssc": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, C.c_size_t, _P]),
    "amx_avg_pool3d_cat": (_I, [_P, _I, C.c_float, _P, _I, C.c_float, _I, _I, _I, _I, _P, _P]),
    "amx_box_filter3d": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "amx_correlate_scratch_bytes": (C.c_size_t, [_I, _I, _I, _I]),
    "amx_correlate_ssd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, C.c_size_t, _P]),
}

_lib = None


AMX_ERR_INVALID, AMX_ERR_SHAPE, AMX_ERR_NOT_LOADED, AMX_ERR_WORKSPACE, AMX_ERR_HIP, AMX_ERR_OVERFLOW = -1, -2, -3, -4, -5, -6


class AmxError(RuntimeError):
    """A non-zero status of the C ABI; ``code`` is the status (include/anatomix_amd.h amx_status)."""

    def __init__(self, *args):
        # (code, message) or just a message
        self.code = args[0] if args and isinstance(args[0], int) else None
        super().__init__(*(args[1:] if self.code is not None else args))


class AmxEnvelopeError(AmxError):
    """The configuration lies outside what an engine entry covers (AMX_ERR_INVALID / AMX_ERR_SHAPE at create time)."""


class AmxOverflowError(AmxError, FloatingPointError):
    """AMX_ERR_OVERFLOW: f16 / f16x2 storage met a value outside the f16 range (include/anatomix_amd.h)."""


def load():
    """dlopen the in-tree library and declare prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AmxError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; "
            "g.build()' or make -C anatomix_amd/csrc).  There is no CPU fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int):
    if status != 0:
        cls = AmxOverflowError if status == AMX_ERR_OVERFLOW else AmxError
        raise cls(int(status), f"anatomix_amd error {status}: {load().amx_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
