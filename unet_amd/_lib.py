"""ctypes binding of libunet_hip.so (the C ABI declared in include/unet_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C unet_amd/csrc``.
There is no fallback: if the shared object is missing or a symbol is absent the first
kernel call raises, so a GPU run can never silently use another code path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UNET_HIP_LIB") or os.path.join(_HERE, "libunet_hip.so")   # override: A/B runs of two builds

MAX_TAPS = 32
BF16, F32 = 0, 1
ACT_IDS = {"none": 0, "relu": 1, "leaky_relu": 2, "gelu": 3, "silu": 4, "mish": 5}


REDUCE_MAX_JOBS = 8       # include/unet_hip.h UNET_REDUCE_MAX_JOBS


class ReduceJob(C.Structure):
    """include/unet_hip.h UnetReduceJob"""
    _fields_ = [("src", C.c_void_p), ("out", C.c_void_p), ("units", C.c_int32), ("rows", C.c_int32), ("c", C.c_int32)]


class PackJob(C.Structure):
    """Mirror of UnetPackJob (include/unet_hip.h)."""
    _fields_ = [
        ("w", C.c_void_p), ("tap_stride", C.c_int64), ("row_stride", C.c_int64), ("col_stride", C.c_int64),
        ("n_taps", C.c_int32), ("has_tap_src", C.c_int32), ("tap_src", C.c_int32 * MAX_TAPS),
        ("row_lo", C.c_int32), ("rows", C.c_int32), ("col_lo", C.c_int32), ("c1", C.c_int32), ("c2", C.c_int32),
        ("packed", C.c_void_p),
    ]


PACK_MAX_JOBS = 4
OPT_MAX_TENSORS = 256


class ConvDesc(C.Structure):
    """Mirror of UnetConvDesc (include/unet_hip.h)."""
    _fields_ = [
        ("dtype", C.c_int32), ("n", C.c_int32), ("l_in", C.c_int32),
        ("c1", C.c_int32), ("c2", C.c_int32),
        ("x1_ld", C.c_int32), ("x2_ld", C.c_int32),
        ("x1_ns", C.c_int64), ("x2_ns", C.c_int64),
        ("l_out", C.c_int32), ("c_out", C.c_int32),
        ("y_ld", C.c_int32), ("y_ns", C.c_int64),
        ("res_ld", C.c_int32), ("res_ns", C.c_int64),
        ("stride", C.c_int32), ("up", C.c_int32), ("n_taps", C.c_int32),
        ("tap_off", C.c_int32 * MAX_TAPS), ("tap_phase", C.c_int32 * MAX_TAPS),
        ("act", C.c_int32),
    ]


class SmallLinear(C.Structure):
    """Mirror of UnetSmallLinear (include/unet_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("b", C.c_void_p), ("y", C.c_void_p), ("pre", C.c_void_p),
                ("part", C.c_void_p), ("n_in", C.c_int32), ("n_out", C.c_int32), ("act", C.c_int32), ("scale", C.c_float),
                ("dy", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("dx_part", C.c_void_p),
                ("dx_slot0", C.c_int32)]


SMALL_MAX = 32
SMALL_ACT = {"none": 0, "gelu": 1, "sigmoid": 2}

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_row_args = [_i32, _i32, _i32, _vp]  # n, rows, c, stream

_SIGNATURES = {
    "unet_abi_version": (C.c_int, []),
    "unet_last_error": (C.c_char_p, []),
    "unet_conv1d_packed_bytes": (C.c_size_t, [_i32, _i32, _i32, _i32, _i32]),
    "unet_conv1d_pack": (C.c_int, [_i32, _vp, _i64, _i64, _i64, _i32, C.POINTER(_i32), _i32, _i32, _i32,
                                   _i32, _i32, _vp, _vp]),
    "unet_conv1d_pack_multi": (C.c_int, [_i32, C.POINTER(PackJob), _i32, _vp]),
    "unet_opt_blocks": (_i32, [_i64]),
    "unet_grad_sqnorm": (C.c_int, [C.POINTER(C.c_void_p), _i32, _vp, _vp, _i64, _vp, _vp, _vp]),
    "unet_clip_adamw_prepare": (C.c_int, [_vp, _i32, C.c_float, _vp, _vp, _vp]),
    "unet_clip_adamw_apply": (C.c_int, [C.POINTER(C.c_void_p), _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, C.c_float,
                                         C.c_float, C.c_float, C.c_float, _vp]),
    "unet_conv1d_fwd": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "unet_conv1d_stats_max_parts": (_i32, [C.POINTER(ConvDesc)]),
    "unet_conv1d_fwd_gated": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "unet_conv1d_fwd_stats": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32), _vp]),
    "unet_conv1d_last_variant": (C.c_int, []),
    "unet_conv1d_wgrad_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "unet_conv1d_wgrad": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "unet_conv1d_wgrad_noalloc_max_bytes": (C.c_size_t, []),
    "unet_conv1d_wgrad_noalloc": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    "unet_parts": (_i32, [_i32]),
    "unet_ncl_moments": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "unet_ncl_to_nlc": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "unet_nlc_to_ncl": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "unet_rowstats": (C.c_int, [_i32, _vp, _vp, _vp] + _row_args),
    "unet_colsum": (C.c_int, [_i32, _vp, _vp] + _row_args),
    "unet_affine_act_fwd": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _vp, _vp] + _row_args),
    "unet_affine_act_bwd_reduce": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp] + _row_args),
    "unet_affine_act_bwd_apply": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp] + _row_args),
    "unet_affine_act_bwd_apply_sum": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp] + _row_args),
    "unet_affine_pool": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _vp] + _row_args),
    "unet_affine_act_scale_fwd": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _vp, _vp] + _row_args),
    "unet_affine_act_se_bwd_reduce": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _i32, _vp] + _row_args),
    "unet_se_combine": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "unet_affine_act_se_bwd_apply": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp] + _row_args),
    "unet_bn_finalize_fwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "unet_bn_finalize_bwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "unet_reduce_rows": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "unet_reduce_parts": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "unet_reduce_multi": (C.c_int, [C.POINTER(ReduceJob), _i32, _vp]),
    "unet_scale_fwd": (C.c_int, [_i32, _vp, _vp, _vp] + _row_args),
    "unet_scale_bwd": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp] + _row_args),
    "unet_film_softmax_fwd": (C.c_int, [_i32, _vp, _vp, _vp, _vp] + _row_args),
    "unet_film_softmax_bwd": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp] + _row_args),
    "unet_gated_add_fwd": (C.c_int, [_i32, _vp, _vp, _vp, _vp] + _row_args),
    "unet_gated_add_bwd": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp] + _row_args),
    "unet_merge_taps_parts": (_i32, [_i32, _i32]),
    "unet_merge_taps_fwd": (C.c_int, [C.POINTER(_vp), _vp, _i32, _i32, _i32, _i32, C.POINTER(_i32), _i32, _vp, _vp]),
    "unet_merge_taps_bwd": (C.c_int, [C.POINTER(_vp), _vp, _i32, _i32, _i32, _i32, C.POINTER(_i32), _i32, _vp,
                                      C.POINTER(_vp), _vp, _vp]),
    "unet_merge_branches_fwd": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), _vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(_i32),
                                          _i32, _vp, _vp, _vp, _vp]),
    "unet_merge_branches_bwd": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), _vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(_i32),
                                          _i32, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _vp]),
    "unet_small_linear_fwd": (C.c_int, [C.POINTER(SmallLinear), _i32, _i32, _vp]),
    "unet_small_linear_bwd": (C.c_int, [C.POINTER(SmallLinear), _i32, _i32, _vp]),
    "unet_small_linear_tiles": (_i32, [_i32]),
    "unet_small_linear_splits": (_i32, [_i32]),
    "unet_pair_moments": (C.c_int, [_i32, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _vp]),
    "unet_sum_rows_f64": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "unet_l1_crop_bwd": (C.c_int, [_i32, _i32, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _f32, _vp, _vp]),
    "unet_row_zscore": (C.c_int, [_i32, _i32, _vp, _vp, _i64, _i32, _f32, _i32, _i32, _vp]),
    "unet_welch_psd": (C.c_int, [_i32, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp]),
    "unet_psd_db_stats": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    "unet_band_power": (C.c_int, [_vp, _i64, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32), _i32, _f32, _vp, _vp]),
    "unet_gather_windows": (C.c_int, [_i32, _i32, _vp, _i64, _i64, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "unet_noise_augment": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _f32, _vp, _vp, _f32, _f32, _i32, _i32, _i32,
                                     _vp, _vp]),
    "unet_cfc_attn_fwd": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp]),
    "unet_cfc_attn_bwd_partials": (C.c_size_t, [_i32, _i32, _i32]),
    "unet_cfc_attn_bwd": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp,
                                    _vp, _vp, _vp]),
    "unet_band_log_power": (C.c_int, [_vp, _vp, _i64, _i32, _i32, C.c_float, _vp, _vp]),
    "unet_overlap_add": (C.c_int, [_i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _vp, _vp, _i32, _i32, _i32, _vp]),
}

_lib = None


def declared_symbols():
    """Every entry point include/unet_hip.h declares (tests check the .so exports them all)."""
    return sorted(_SIGNATURES)


def load():
    """Load libunet_hip.so and bind every declared symbol; raise loudly on any problem."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C unet_amd/csrc`). unet_amd has no non-HIP fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.unet_abi_version() != 1:
        raise RuntimeError(f"libunet_hip ABI version {lib.unet_abi_version()} != 1")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {load().unet_last_error().decode()}")
