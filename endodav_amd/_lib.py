"""ctypes binding of ``libendodav_hip.so`` (C ABI: ``include/endodav_hip.h``).

The library is built in-tree by ``make`` / ``__graft_entry__.build()``.  There is no fallback:
if the shared object is missing or a symbol is absent, importing the binding raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EDV_LIB_PATH") or os.path.join(_HERE, "lib", "libendodav_hip.so")  # override: experiments only
ABI_VERSION = 10

LORA_TYPES = {"none": 0, "lora": 1, "dvlora": 2, "ssb": 3, "dash": 4}
ACT_NONE, ACT_GELU, ACT_RELU, ACT_SIGMOID, ACT_SIGMOID_NEG = 0, 1, 2, 3, 4


class EdvConfig(C.Structure):
    """``struct edv_config`` — field order must match the header."""

    _fields_ = [
        ("abi_version", C.c_int32),
        ("embed_dim", C.c_int32),
        ("depth", C.c_int32),
        ("num_heads", C.c_int32),
        ("taps", C.c_int32 * 4),
        ("features", C.c_int32),
        ("out_channels", C.c_int32 * 4),
        ("image_h", C.c_int32),
        ("image_w", C.c_int32),
        ("num_frames", C.c_int32),
        ("pos_tokens", C.c_int32),
        ("lora_type", C.c_int32),
        ("lora_rank", C.c_int32),
        ("include_cls_token", C.c_int32),
        ("conv_head", C.c_int32),
        ("inv_sigmoid", C.c_int32),
        ("out_sigmoid", C.c_int32),
        ("temporal_lora", C.c_int32),
        ("dash_active", C.c_int32),
        ("use_clstoken", C.c_int32),
        ("residual_mask", C.c_uint32),
        ("use_bn", C.c_int32),
        ("pe_rope", C.c_int32),
    ]


_fp = C.c_void_p  # device pointers travel as integers
_i32, _i64, _f32, _f64 = C.c_int32, C.c_int64, C.c_float, C.c_double

# name -> (restype, argtypes); every symbol declared in include/endodav_hip.h
SIGNATURES = {
    "edv_abi_version": (C.c_int, []),
    "edv_last_error": (C.c_char_p, []),
    "edv_create": (C.c_int, [C.POINTER(EdvConfig), C.POINTER(C.c_void_p)]),
    "edv_destroy": (C.c_int, [C.c_void_p]),
    "edv_bind_param": (C.c_int, [C.c_void_p, C.c_char_p, _fp, C.POINTER(_i64), _i32]),
    "edv_prepare": (C.c_int, [C.c_void_p, C.c_void_p]),
    "edv_refresh_lora": (C.c_int, [C.c_void_p, C.c_void_p]),
    "edv_forward": (C.c_int, [C.c_void_p, _fp, _i32, _i32, _i32, _i32, C.POINTER(C.c_void_p), C.c_void_p]),
    "edv_output_shape": (C.c_int, [C.c_void_p, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "edv_stage_copy": (C.c_int, [C.c_void_p, C.c_char_p, _fp, C.POINTER(C.c_size_t), C.c_void_p]),
    "edv_set_capture": (C.c_int, [C.c_void_p, C.c_int]),
    "edv_profile_enable": (C.c_int, [C.c_void_p, C.c_uint32]),
    "edv_set_encoder_streams": (C.c_int, [C.c_void_p, _i32]),
    "edv_set_products": (C.c_int, [C.c_void_p, _i32, C.c_void_p]),
    "edv_get_products": (C.c_int, [C.c_void_p]),
    "edv_profile_set_mask": (C.c_int, [C.c_void_p, C.c_uint32]),
    "edv_profile_work": (C.c_int, [C.c_void_p, _i32, C.POINTER(_f64), C.POINTER(_f64)]),
    "edv_profile_read": (C.c_int, [C.c_void_p, _i32, C.POINTER(_i32), C.POINTER(_f64)]),
    "edv_device_bytes": (C.c_size_t, [C.c_void_p]),
    "edv_last_launch_count": (C.c_int, [C.c_void_p]),
    "edv_layernorm": (C.c_int, [_fp, _fp, _fp, _fp, _i64, _i32, _f32, _fp, _i32, _i32, C.c_void_p]),
    "edv_gemm_workspace": (C.c_size_t, []),
    "edv_gemm_x6_planes_bytes": (C.c_size_t, [_i32, _i32]),
    "edv_gemm_x6_split": (C.c_int, [_fp, C.c_void_p, _i32, _i32, C.c_void_p]),
    "edv_gemm_x6": (C.c_int, [_fp, C.c_void_p, _fp, _i64, _i32, _i32, _fp, _i32, _fp, _fp, _fp, C.c_size_t, C.c_void_p]),
    "edv_gemm": (C.c_int, [_fp, _fp, _fp, _i64, _i32, _i32, _fp, _i32, _fp, _fp, _fp, C.c_size_t, C.c_void_p]),
    "edv_pack_geglu": (C.c_int, [_fp, _fp, _fp, _fp, _i32, _i32, C.c_void_p]),
    "edv_gemm_geglu": (C.c_int, [_fp, _fp, _fp, _fp, _i64, _i32, _i32, C.c_void_p]),
    "edv_conv3x3": (C.c_int, [_fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _fp, _fp, C.c_void_p]),
    "edv_conv3x3_ws": (C.c_int, [_fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _fp, _fp, _fp, C.c_size_t, C.c_void_p]),
    "edv_pack_conv3x3": (C.c_int, [_fp, _fp, _i32, _i32, C.c_void_p]),
    "edv_conv_transpose": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, C.c_void_p]),
    "edv_attn_spatial_workspace": (C.c_size_t, [_i32, _i32, _i32]),
    "edv_attn_spatial": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _fp, C.c_size_t, _fp, C.c_void_p]),
    "edv_set_train": (C.c_int, [C.c_void_p, _i32]),
    "edv_set_grad_scope": (C.c_int, [C.c_void_p, _i32, _i32, _i32, _i32]),
    "edv_generation": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "edv_backward": (C.c_int, [C.c_void_p, C.c_uint64, _fp, C.POINTER(C.c_void_p), C.c_void_p]),
    "edv_grad_bind_flat": (C.c_int, [C.c_void_p, _i32, C.POINTER(C.c_char_p), C.POINTER(_i64), _fp, _i64, C.POINTER(_i64)]),
    "edv_grad": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(_i64)]),
    "edv_grad_copy": (C.c_int, [C.c_void_p, C.c_char_p, _fp, _i64, C.c_void_p]),
    "edv_attn_spatial_x6_workspace": (C.c_size_t, [_i32, _i32, _i32]),
    "edv_attn_spatial_x6": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _fp, C.c_size_t, C.c_void_p]),
    "edv_attn_spatial_bwd_workspace": (C.c_size_t, [_i32, _i32, _i32]),
    "edv_attn_spatial_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _fp, C.c_size_t, C.c_void_p]),
    "edv_layernorm_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _i64, _i32, C.c_float, _i32, C.c_void_p]),
    "edv_ew_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _i64, _i32, C.c_void_p]),
    "edv_geglu_bwd": (C.c_int, [_fp, _fp, _fp, _i64, _i32, C.c_void_p]),
    "edv_transpose_scale": (C.c_int, [_fp, _fp, _fp, _i32, _i32, C.c_void_p]),
    "edv_lora_grads_workspace": (C.c_size_t, [_i64, _i32, _i32, _i32]),
    "edv_lora_grads": (C.c_int, [_fp, _fp, _i64, _i32, _i32, _i32, _fp, _fp, _fp, _fp, C.c_float, _fp, _fp, C.c_size_t, _fp, _fp, _fp, _fp, C.c_void_p]),
    "edv_bilinear_bwd": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, C.c_void_p]),
    "edv_dot_channels_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _i64, _i32, _i32, C.c_void_p]),
    "edv_conv3x3_wgrad_workspace": (C.c_size_t, [_i32, _i32, _i32, _i32, _i32]),
    "edv_conv3x3_wgrad": (C.c_int, [_fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, _fp, C.c_size_t, _i32, C.c_void_p]),
    "edv_colsum_workspace": (C.c_size_t, [_i32]),
    "edv_colsum_rows": (C.c_int, [_fp, _fp, _i64, _i32, _fp, C.c_size_t, _fp, _i32, C.c_void_p]),
    "edv_groupnorm_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, C.c_void_p]),
    "edv_attn_temporal_bwd": (C.c_int, [_fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, C.c_void_p]),
    "edv_pack_conv3x3_bwd": (C.c_int, [_fp, _fp, _i32, _i32, C.c_void_p]),
    "edv_conv3x3_s2_bwd": (C.c_int, [_fp, _fp, _fp, _i32, _i32, _i32, _i32, _i32, C.c_void_p]),
    "edv_dilate2": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, C.c_void_p]),
    "edv_pixel_unshuffle": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, C.c_void_p]),
    "edv_attn_temporal": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, C.c_void_p]),
    "edv_groupnorm_workspace": (C.c_size_t, [_i32, _i32, _i32]),
    "edv_groupnorm": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _f32, _fp, C.c_size_t, C.c_void_p]),
    "edv_geglu": (C.c_int, [_fp, _fp, _i64, _i32, C.c_void_p]),
    "edv_rope_qk": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, C.c_void_p]),
    "edv_trainer_loss_workspace": (C.c_size_t, [_i32, _i32, _i32]),
    "edv_trainer_loss": (C.c_int, [C.c_void_p, _i32, _i32, _i32, C.c_void_p, _fp, C.c_void_p, _fp, C.c_size_t, C.c_void_p]),
    "edv_debug_fill_lds": (C.c_int, [C.c_float, C.c_void_p]),
    "edv_bilinear": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, _i32, C.c_void_p]),
    "edv_dot_channels": (C.c_int, [_fp, _fp, _fp, _fp, _i64, _i32, _i32, C.c_void_p]),
    "edv_patchify": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, C.c_void_p]),
    "edv_bicubic_pos": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _f64, _f64, C.c_void_p]),
    "edv_resize_bicubic": (C.c_int, [_fp, _fp, _i32, _i32, _i32, _i32, _i32, C.c_void_p]),
    "edv_photometric_loss_workspace": (C.c_size_t, [_i32, _i32, _i32, _i32]),
    "edv_photometric_loss": (C.c_int, [_fp, C.POINTER(C.c_void_p), C.POINTER(_i32), C.POINTER(_i32), _i32, _i32, _i32, _i32, _fp, _fp, _fp, _fp, _f32, _f32, _f32,
                                       _fp, C.POINTER(C.c_void_p), _fp, C.c_size_t, C.c_void_p]),
    "edv_fold_lora": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _f32, _fp, _i32, _i32, _i32, C.c_void_p]),
}


# "linear_encoder" (the F.linear launches of the encoder blocks) is recorded whenever "linear" is enabled and read separately;
# "linear" then holds the remaining F.linear / 1x1-conv launches (patch embed, DPT head)
KERNEL_CLASSES = {"linear": 0, "conv3x3": 1, "attn_spatial": 2, "attn_temporal": 3, "layernorm": 4, "other": 5, "linear_encoder": 6,
                  "groupnorm": 7, "bilinear": 8, "geglu": 9, "dot_channels": 10, "patchify": 11, "attn_spatial_bwd": 12}
HBM_CLASSES = ("layernorm", "groupnorm", "bilinear", "geglu", "dot_channels", "patchify")  # bandwidth-bound kernels (bench.py roofline_hbm)


class EdvError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library once and type every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EdvError(
            f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950) or "
            "`python -c 'import __graft_entry__ as g; g.build()'`.  There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is absent
        fn.restype = res
        fn.argtypes = args
    got = lib.edv_abi_version()
    if got != ABI_VERSION:
        raise EdvError(f"libendodav_hip ABI {got} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().edv_last_error()
        raise EdvError(f"{what}: {msg.decode() if msg else 'unknown error'}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (0 for None)."""
    return 0 if t is None else t.data_ptr()


class TrainerLossInputs(C.Structure):
    """edv_trainer_loss_inputs (include/endodav_hip.h)."""
    _fields_ = [("color", C.c_void_p * 4), ("color_nb", C.c_void_p * 2), ("K", C.c_void_p), ("invK", C.c_void_p), ("T", C.c_void_p * 2),
                ("refined", (C.c_void_p * 2) * 4), ("registration", (C.c_void_p * 2) * 4), ("transform", (C.c_void_p * 2) * 4), ("mask", C.c_void_p * 2),
                ("position", (C.c_void_p * 2) * 4), ("disp", C.c_void_p * 4), ("disp_h", C.c_int32 * 4), ("disp_w", C.c_int32 * 4)]


class TrainerLossWeights(C.Structure):
    _fields_ = [("disparity_smoothness", C.c_float), ("transform_constraint", C.c_float), ("transform_smoothness", C.c_float), ("depth_reproj", C.c_float),
                ("depth_flow", C.c_float), ("tune_temporal", C.c_int32), ("min_depth", C.c_float), ("max_depth", C.c_float)]


class TrainerLossGrads(C.Structure):
    _fields_ = [("disp", C.c_void_p * 4), ("refined", (C.c_void_p * 2) * 4), ("transform", (C.c_void_p * 2) * 4), ("K", C.c_void_p), ("invK", C.c_void_p),
                ("T", C.c_void_p * 2)]


def stream_ptr(device=None) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream
