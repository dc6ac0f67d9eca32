"""ctypes binding of libidb_kernels.so (the C ABI declared in include/idb_kernels.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C faceposegenerator_amd/csrc``.
There is no fallback: if the shared object is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# IDB_LIB selects another build of the same ABI (measurement tools: the profiling library libidb_kernels_prof.so, `make prof`)
LIB_PATH = os.environ.get("IDB_LIB") or os.path.join(_HERE, "libidb_kernels.so")

IDB_BF16, IDB_F16, IDB_F32 = 0, 1, 2
IDB_MAX_SRC = 4

# every symbol include/idb_kernels.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "idb_version", "idb_launch_count", "idb_last_error", "idb_device_check",
    "idb_gemm_workspace_bytes", "idb_gemm_plan", "idb_gemm_row_stats_tiles", "idb_gemm_folds_layernorm", "idb_gemm_fuses_groupnorm", "idb_gemm_emits_gn_partials", "idb_gemm",
    "idb_pack_conv_weight", "idb_pack_matrix", "idb_tiled_weight_bytes", "idb_tile_weight", "idb_lora_merge", "idb_lora_merge_scaled", "idb_pack_matrix_scaled", "idb_ln_fold_vectors",
    "idb_groupnorm_workspace_bytes", "idb_groupnorm", "idb_layernorm", "idb_groupnorm_stats",
    "idb_attention", "idb_embed_tokens", "idb_softmax_rows",
    "idb_timestep_sinusoid", "idb_linear_f32", "idb_conv_in",
    "idb_cfg_ddpm_step", "idb_postprocess",
    "idb_nhwc_to_nchw_f32", "idb_f32_nhwc_to_nchw", "idb_cast_f32", "idb_vae_sample", "idb_warp_affine_u8",
    "idb_crop_resize_area_u8", "idb_conv2d_f32", "idb_maxpool2d_f32", "idb_softmax_pairs_f32", "idb_nms_mask",
    "idb_quantize_fp8", "idb_pack_weight_fp8", "idb_gemm_fp8", "idb_groupnorm_fp8",
]


class GemmSrc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("channels", C.c_int32), ("taps", C.c_int32),
                ("in_h", C.c_int32), ("in_w", C.c_int32), ("upsample", C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("batch", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32),
                ("stride", C.c_int32), ("n", C.c_int32), ("nsrc", C.c_int32),
                ("src", GemmSrc * IDB_MAX_SRC),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("sample_bias", C.c_void_p),
                ("sample_bias_ld", C.c_int32), ("residual", C.c_void_p), ("geglu", C.c_int32),
                ("out", C.c_void_p), ("out_dtype", C.c_int32), ("out_ld", C.c_int32),
                ("split_k", C.c_int32), ("tile", C.c_int32), ("out_scale", C.c_float), ("flags", C.c_int32), ("act", C.c_int32),
                ("counters", C.c_void_p), ("counters_len", C.c_int32), ("gn_partials", C.c_void_p), ("gn_groups", C.c_int32),
                ("row_stats_out", C.c_void_p), ("ln_stats", C.c_void_p), ("ln_tiles", C.c_int32), ("ln_u", C.c_void_p),
                ("ln_v", C.c_void_p), ("ln_eps", C.c_float), ("pad_mode", C.c_int32), ("w_layout", C.c_int32),
                ("w_groups", C.c_int32), ("w_group_rows", C.c_int32), ("w_group_stride", C.c_int64),
                ("gn_in_partials", C.c_void_p), ("gn_in_chunks", C.c_int32), ("gn_in_groups", C.c_int32), ("gn_in_nsrc", C.c_int32),
                ("gn_in_silu", C.c_int32), ("gn_in_eps", C.c_float), ("gn_in_gamma", C.c_void_p), ("gn_in_beta", C.c_void_p)]


class GemmFp8Desc(C.Structure):
    _fields_ = [("out_dtype", C.c_int32), ("batch", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32), ("stride", C.c_int32),
                ("n", C.c_int32), ("x", C.c_void_p), ("channels", C.c_int32), ("taps", C.c_int32), ("in_h", C.c_int32),
                ("in_w", C.c_int32), ("upsample", C.c_int32), ("x_scale", C.c_float), ("w", C.c_void_p), ("w_scale", C.c_void_p),
                ("bias", C.c_void_p), ("sample_bias", C.c_void_p), ("sample_bias_ld", C.c_int32), ("residual", C.c_void_p),
                ("out", C.c_void_p), ("out_ld", C.c_int32), ("gn_partials", C.c_void_p), ("gn_groups", C.c_int32)]


class IdbError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared object (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise IdbError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C faceposegenerator_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
    sig = {
        "idb_version": (C.c_int, []),
        "idb_launch_count": (C.c_uint64, []),
        "idb_last_error": (C.c_char_p, []),
        "idb_device_check": (C.c_int, [C.c_int]),
        "idb_gemm_workspace_bytes": (sz, [C.POINTER(GemmDesc)]),
        "idb_gemm_plan": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "idb_gemm": (C.c_int, [C.POINTER(GemmDesc), vp, sz, vp]),
        "idb_gemm_row_stats_tiles": (i32, [C.POINTER(GemmDesc)]),
        "idb_gemm_folds_layernorm": (i32, [C.POINTER(GemmDesc)]),
        "idb_gemm_fuses_groupnorm": (i32, [C.POINTER(GemmDesc)]),
        "idb_gemm_emits_gn_partials": (i32, [C.POINTER(GemmDesc), i32]),
        "idb_lora_merge_scaled": (C.c_int, [vp, vp, vp, vp, i64, i64, i32, f32, vp, i32, vp]),
        "idb_pack_conv_weight": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
        "idb_pack_matrix": (C.c_int, [vp, vp, i64, i64, i32, i32, vp]),
        "idb_pack_matrix_scaled": (C.c_int, [vp, vp, i64, i64, i32, vp, i32, vp]),
        "idb_ln_fold_vectors": (C.c_int, [vp, vp, vp, i32, f32, vp, vp, vp, vp, vp, i64, i64, i32, i32, vp]),
        "idb_tiled_weight_bytes": (sz, [i64, i64]),
        "idb_tile_weight": (C.c_int, [vp, vp, i64, i64, i32, vp]),
        "idb_lora_merge": (C.c_int, [vp, vp, vp, vp, i64, i64, i32, f32, i32, vp]),
        "idb_groupnorm_workspace_bytes": (sz, [i32, i32, i32]),
        "idb_groupnorm": (C.c_int, [vp, i32, vp, i32, i32, i32, i32, f32, vp, vp, i32, vp, i32, vp, sz, vp, i32, vp, i32, vp]),
        "idb_layernorm": (C.c_int, [vp, vp, i64, i32, f32, vp, vp, i32, vp]),
        "idb_groupnorm_stats": (C.c_int, [vp, i32, vp, i32, i32, i32, i32, vp, sz, C.POINTER(i32), i32, vp]),
        "idb_attention": (C.c_int, [vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, f32, i32, i32, vp]),
        "idb_embed_tokens": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
        "idb_softmax_rows": (C.c_int, [vp, i64, i32, i32, vp]),
        "idb_timestep_sinusoid": (C.c_int, [vp, vp, i32, i32, vp]),
        "idb_linear_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
        "idb_conv_in": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, i32, vp]),
        "idb_cfg_ddpm_step": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "idb_postprocess": (C.c_int, [vp, vp, vp, i64, vp]),
        "idb_nhwc_to_nchw_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
        "idb_f32_nhwc_to_nchw": (C.c_int, [vp, vp, i32, i32, i32, vp]),
        "idb_cast_f32": (C.c_int, [vp, vp, i64, i32, vp]),
        "idb_vae_sample": (C.c_int, [vp, vp, f32, vp, vp, vp, i32, i32, i32, vp]),
        "idb_warp_affine_u8": (C.c_int, [vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, vp]),
        "idb_crop_resize_area_u8": (C.c_int, [vp, i32, i32, i32, i32, vp, i32, vp, i32, i32, f32, f32, vp]),
        "idb_conv2d_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
        "idb_maxpool2d_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
        "idb_softmax_pairs_f32": (C.c_int, [vp, vp, i32, i32, vp]),
        "idb_nms_mask": (C.c_int, [vp, vp, i32, C.c_float, i32, i32, vp, vp]),
        "idb_quantize_fp8": (C.c_int, [vp, vp, i64, f32, i32, vp]),
        "idb_pack_weight_fp8": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
        "idb_gemm_fp8": (C.c_int, [C.POINTER(GemmFp8Desc), vp]),
        "idb_groupnorm_fp8": (C.c_int, [vp, i32, vp, i32, i32, i32, i32, f32, vp, vp, i32, vp, f32, i32, vp, sz, vp, i32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().idb_last_error()
        raise IdbError(f"{what or 'idb call'} failed (status {rc}): {msg.decode() if msg else ''}")
