"""ctypes binding of libccnet_hip.so (the C ABI declared in include/ccnet_hip.h).

The library is the only compute path of this package: there is no CPU or eager-PyTorch fallback.
`load()` raises if the shared object is missing, and every op raises `CcnetError` on a non-zero
return code.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
# CCNET_HIP_LIB: load another build of the same ABI (A/B timing of kernel variants); default is the in-tree library
LIB_PATH = Path(os.environ.get("CCNET_HIP_LIB", str(_HERE / "libccnet_hip.so")))

EXPORTS = [
    "ccnet_strerror", "ccnet_version", "ccnet_arch", "ccnet_conv_gemm_bf16", "ccnet_conv_in_small",
    "ccnet_conv_out_small", "ccnet_preprocess_u8_bicubic", "ccnet_groupnorm_workspace_bytes", "ccnet_groupnorm_bf16", "ccnet_layernorm_bf16",
    "ccnet_geglu_bf16", "ccnet_attention_d64_bf16", "ccnet_softmax_rows_bf16", "ccnet_timestep_embedding",
    "ccnet_linear_f32", "ccnet_cfg_scheduler_step", "ccnet_nchw_f32_to_nhwc_bf16", "ccnet_nhwc_bf16_to_nchw_f32",
    "ccnet_to_uint8", "ccnet_prof_enable", "ccnet_prof_collect", "ccnet_pointwise_nchw_f32", "ccnet_axpy_f32",
    "ccnet_set_option", "ccnet_prof_shapes", "ccnet_conv_workspace_bytes", "ccnet_transpose_pad_bf16",
    "ccnet_attention_d64_causal_bf16", "ccnet_embed_tokens_bf16", "ccnet_gelu_bf16", "ccnet_conv_gemm_bf16_noalloc",
    "ccnet_row_stats_bf16", "ccnet_conv_stat_parts",
    # training step (backward kernels)
    "ccnet_im2col_t_bf16", "ccnet_im2col_t_nchw_f32", "ccnet_colsum_bf16", "ccnet_groupnorm_bwd_bf16",
    "ccnet_layernorm_bwd_bf16", "ccnet_softmax_bwd_rows_bf16", "ccnet_geglu_bwd_bf16", "ccnet_silu_bf16", "ccnet_axpby_bf16",
    "ccnet_resample_bwd_bf16", "ccnet_attention_d64_lse_bf16", "ccnet_attention_d64_bwd_bf16",
    "ccnet_attention_d64_prescaled_bf16", "ccnet_ff_geglu_fused_bf16", "ccnet_ff_packed_bytes", "ccnet_noalloc_init",
    "ccnet_xattn_fused_bf16", "ccnet_xattn_packed_bytes", "ccnet_xattn_kv_image_bytes",
]


class CcnetError(RuntimeError):
    pass


class CcnetUnsupported(CcnetError):
    """CCNET_EUNSUP: a configuration the kernels do not cover (callers with another lowering of the same op may take it)."""


class ConvParams(C.Structure):
    """Mirror of `ccnet_conv_params` (include/ccnet_hip.h)."""
    _fields_ = [
        ("x1", C.c_void_p), ("x2", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("rowvec", C.c_void_p),
        ("res1", C.c_void_p), ("res2", C.c_void_p), ("out", C.c_void_p), ("out2", C.c_void_p),
        ("N", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("C1", C.c_int32), ("C2", C.c_int32), ("Cout", C.c_int32), ("R", C.c_int32), ("stride", C.c_int32),
        ("pad", C.c_int32), ("ups", C.c_int32),
        ("ldx1", C.c_int32), ("ldx2", C.c_int32), ("ldw", C.c_int32), ("ldo", C.c_int32), ("ldo2", C.c_int32),
        ("ldres1", C.c_int32), ("ldres2", C.c_int32), ("ld_rowvec", C.c_int32),
        ("act", C.c_int32), ("scale", C.c_float), ("batch", C.c_int32),
        ("sx1", C.c_int64), ("sw", C.c_int64), ("so", C.c_int64), ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64), ("rowstat", C.c_void_p), ("colsum", C.c_void_p),
        ("rowstat_parts", C.c_int32), ("ln_eps", C.c_float), ("stat_out", C.c_void_p),
    ]


_lib = None


def load() -> C.CDLL:
    """Load the HIP library; fail loudly if it has not been built (`__graft_entry__.build()`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise CcnetError(f"{LIB_PATH} not found: build it with `make -C {_HERE / 'csrc'}` "
                         "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
    lib = C.CDLL(str(LIB_PATH))
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    lib.ccnet_strerror.restype = C.c_char_p
    lib.ccnet_strerror.argtypes = [i]
    lib.ccnet_version.restype = i
    lib.ccnet_arch.restype = C.c_char_p
    lib.ccnet_conv_gemm_bf16.argtypes = [C.POINTER(ConvParams), vp]
    lib.ccnet_conv_gemm_bf16_noalloc.argtypes = [C.POINTER(ConvParams), vp]
    lib.ccnet_conv_workspace_bytes.argtypes = [C.POINTER(ConvParams)]
    lib.ccnet_conv_stat_parts.argtypes = [C.POINTER(ConvParams)]
    lib.ccnet_conv_stat_parts.restype = C.c_int
    lib.ccnet_conv_workspace_bytes.restype = sz
    lib.ccnet_conv_in_small.argtypes = [vp, i, i, i, i, i, vp, vp, i, i, i, vp, i, i, vp, i, i, i, vp]
    lib.ccnet_conv_out_small.argtypes = [vp, i, i, i, i, i, vp, vp, i, vp, vp]
    lib.ccnet_groupnorm_workspace_bytes.restype = sz
    lib.ccnet_groupnorm_workspace_bytes.argtypes = [i, i, i, i]
    lib.ccnet_groupnorm_bf16.argtypes = [vp, i, i, vp, i, i, i, i, i, f, vp, vp, i, vp, i, vp, sz, vp]
    lib.ccnet_layernorm_bf16.argtypes = [vp, i, i, i, f, vp, vp, vp, i, vp]
    lib.ccnet_row_stats_bf16.argtypes = [vp, i, i, i, f, vp, vp]
    lib.ccnet_geglu_bf16.argtypes = [vp, i, i, i, vp, i, vp]
    lib.ccnet_attention_d64_bf16.argtypes = [vp, i, vp, i, vp, i, vp, i, i, i, i, i, f, vp]
    lib.ccnet_attention_d64_causal_bf16.argtypes = [vp, i, vp, i, vp, i, vp, i, i, i, i, f, vp]
    lib.ccnet_attention_d64_prescaled_bf16.argtypes = [vp, i, vp, i, vp, i, vp, i, i, i, i, i, vp]
    lib.ccnet_embed_tokens_bf16.argtypes = [vp, i, i, i, vp, vp, i, vp, i, vp]
    lib.ccnet_gelu_bf16.argtypes = [vp, i, i, i, i, vp, i, vp]
    lib.ccnet_softmax_rows_bf16.argtypes = [vp, i, i, i, i, f, vp, i, vp]
    lib.ccnet_transpose_pad_bf16.argtypes = [vp, i, i, i, C.c_int64, vp, i, i, C.c_int64, i, vp]
    lib.ccnet_timestep_embedding.argtypes = [vp, i, i, i, vp, vp, vp]
    lib.ccnet_linear_f32.argtypes = [vp, i, i, i, vp, vp, i, i, i, vp, i, vp, i, vp]
    lib.ccnet_cfg_scheduler_step.argtypes = [vp, i, f, vp, vp, vp, vp, i, i, vp]
    lib.ccnet_nchw_f32_to_nhwc_bf16.argtypes = [vp, i, i, i, vp, i, vp]
    lib.ccnet_nhwc_bf16_to_nchw_f32.argtypes = [vp, i, i, i, i, vp, vp]
    lib.ccnet_to_uint8.argtypes = [vp, i, i, i, i, vp, vp]
    lib.ccnet_preprocess_u8_bicubic.argtypes = [vp, i, i, i, i, vp, i, i, vp]
    lib.ccnet_pointwise_nchw_f32.argtypes = [vp, i, i, i, vp, vp, i, i, vp, vp]
    lib.ccnet_axpy_f32.argtypes = [vp, vp, vp, C.c_longlong, vp, vp]
    lib.ccnet_set_option.argtypes = [i, i]
    lib.ccnet_prof_enable.argtypes = [i]
    lib.ccnet_prof_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.ccnet_prof_shapes.argtypes = [C.POINTER(C.c_double), i]
    i64 = C.c_int64
    lib.ccnet_im2col_t_bf16.argtypes = [vp, i, i, vp, i, i, i, i, i, i, i, i, i, i, i, vp, i64, vp]
    lib.ccnet_im2col_t_nchw_f32.argtypes = [vp, i, i, i, i, i, i, vp, i64, i, vp]
    lib.ccnet_colsum_bf16.argtypes = [vp, i, i, i, i, vp, i, vp, i, vp, vp, vp]
    lib.ccnet_groupnorm_bwd_bf16.argtypes = [vp, i, i, vp, i, i, vp, i, i, i, i, f, vp, vp, i, vp, i, vp, i, vp, vp]
    lib.ccnet_layernorm_bwd_bf16.argtypes = [vp, i, vp, i, i, i, f, vp, vp, i, vp, vp]
    lib.ccnet_softmax_bwd_rows_bf16.argtypes = [vp, vp, i, i, i, i, f, vp]
    lib.ccnet_geglu_bwd_bf16.argtypes = [vp, i, vp, i, i, i, vp, i, vp]
    lib.ccnet_silu_bf16.argtypes = [vp, i, vp, i, i, i, i, vp, i, vp]
    lib.ccnet_axpby_bf16.argtypes = [vp, i, vp, i, i, i, f, f, vp]
    lib.ccnet_resample_bwd_bf16.argtypes = [vp, i, vp, i, i, i, i, i, i, i, i, vp]
    lib.ccnet_attention_d64_lse_bf16.argtypes = [vp, i, vp, i, vp, i, vp, i, i, i, i, i, f, vp, vp]
    lib.ccnet_attention_d64_bwd_bf16.argtypes = [vp, i, vp, i, vp, i, vp, i, vp, i, vp, vp, vp, i, vp, i, vp, i, i, i, i, i, f, vp]
    lib.ccnet_ff_geglu_fused_bf16.argtypes = [vp, vp]
    lib.ccnet_ff_packed_bytes.argtypes = [i, i]
    lib.ccnet_ff_packed_bytes.restype = C.c_int64
    lib.ccnet_xattn_fused_bf16.argtypes = [vp, vp]
    lib.ccnet_xattn_packed_bytes.argtypes = [i, i]
    lib.ccnet_xattn_packed_bytes.restype = C.c_int64
    lib.ccnet_xattn_kv_image_bytes.argtypes = [i, i]
    lib.ccnet_xattn_kv_image_bytes.restype = C.c_int64
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int:  # default
            fn.restype = C.c_int
    # tuning knobs for A/B timing (tools/, bench.py): CCNET_HIP_OPTIONS="key=value,key=value" -> ccnet_set_option
    for kv in filter(None, os.environ.get("CCNET_HIP_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        if lib.ccnet_set_option(int(k), int(v)) != 0:
            raise CcnetError(f"CCNET_HIP_OPTIONS: unknown option {kv!r}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise (CcnetUnsupported if rc == -2 else CcnetError)(f"{what}: {load().ccnet_strerror(rc).decode()} (code {rc})")
