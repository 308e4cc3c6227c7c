"""ctypes binding of libfyc_hip.so (include/fyc.h).  Fails loudly: there is no CPU fallback.

The structures below mirror include/fyc.h field for field; tests/test_abi.py checks that every
symbol the header declares is exported and that the struct sizes agree with a C compile of the header.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FYC_LIB_PATH") or os.path.join(HERE, "libfyc_hip.so")   # FYC_LIB_PATH: A/B builds

FYC_F32, FYC_BF16, FYC_F16 = 0, 1, 2
GEMM_PLAIN, GEMM_CONV3X3, GEMM_CONV3X3_UP2 = 0, 1, 2
EPI_LINEAR, EPI_GEGLU, EPI_HEADS = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_QUICK_GELU = 0, 1, 2

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [("a", vp), ("a2", vp), ("w", vp), ("bias", vp), ("rowbias", vp), ("residual", vp), ("out", vp),
                ("seg_out", vp * 3), ("seg_transposed", i32 * 3), ("seg_ld", i32 * 3),
                ("M", i32), ("N", i32), ("K", i32),
                ("lda", i32), ("ldw", i32), ("ldo", i32), ("ldr", i32), ("ldrb", i32), ("k_split", i32), ("lda2", i32),
                ("stride_a", i64), ("stride_w", i64), ("stride_o", i64),
                ("batch", i32), ("mode", i32), ("epilogue", i32),
                ("Hout", i32), ("Wout", i32), ("Hin", i32), ("Win", i32), ("Cin", i32), ("conv_stride", i32), ("conv_pad", i32),
                ("rows_per_batch", i32), ("seg_cols", i32), ("heads", i32), ("tokens", i32),
                ("out_scale", f32), ("dtype", i32), ("tile", i32), ("act", i32), ("ln_stats", vp), ("ln_colsum", vp),
                ("ln_nparts", i32), ("ln_eps", f32), ("chan_parts", vp), ("cs_rows", i32), ("row_parts", vp), ("row_nparts", i32), ("workspace", vp), ("workspace_bytes", i64)]


class AttnArgs(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("vt", vp), ("o", vp),
                ("batch", i32), ("heads", i32), ("n_q", i32), ("n_k", i32), ("d", i32),
                ("ldo", i32), ("ldvt", i32), ("kv_batch_div", i32), ("o_accumulate", i32),
                ("scale", f32), ("o_scale", f32), ("dtype", i32)]


class TAttnArgs(C.Structure):
    _fields_ = [("qkv", vp), ("o", vp), ("clips", i32), ("frames", i32), ("pixels", i32), ("heads", i32), ("d", i32),
                ("scale", f32), ("dtype", i32)]


class GnStatsArgs(C.Structure):
    _fields_ = [("x", vp), ("stats", vp), ("rows", i32), ("C", i32), ("groups", i32), ("rows_per_sample", i32), ("dtype", i32),
                ("pad_", i32), ("workspace", vp), ("workspace_bytes", i64)]


class GnApplyArgs(C.Structure):
    _fields_ = [("x", vp), ("stats", vp), ("gamma", vp), ("beta", vp), ("y", vp),
                ("rows", i32), ("C", i32), ("groups", i32), ("rows_per_sample", i32), ("eps", f32), ("silu", i32), ("dtype", i32)]


class ChanStatsReduceArgs(C.Structure):
    _fields_ = [("parts", vp), ("cs", vp), ("rows", i32), ("N", i32), ("cs_rows", i32), ("tile_rows", i32), ("slots", i32), ("out_rows", i32)]


class GnApplyCsArgs(C.Structure):
    _fields_ = [("x1", vp), ("cs1", vp), ("x2", vp), ("cs2", vp), ("gamma", vp), ("beta", vp), ("y", vp),
                ("C1", i32), ("C2", i32), ("rows", i32), ("groups", i32), ("rows_per_sample", i32), ("eps", f32), ("silu", i32),
                ("dtype", i32), ("cs_rows", i32)]


class LayerNormArgs(C.Structure):
    _fields_ = [("x", vp), ("gamma", vp), ("beta", vp), ("pe", vp), ("y", vp),
                ("rows", i32), ("C", i32), ("eps", f32), ("pe_div", i32), ("pe_rows", i32), ("dtype", i32)]


class RowStatsArgs(C.Structure):
    _fields_ = [("x", vp), ("stats", vp), ("rows", i32), ("C", i32), ("eps", f32), ("dtype", i32)]


class SoftmaxArgs(C.Structure):
    _fields_ = [("x", vp), ("rows", i64), ("cols", i32), ("ld", i32), ("dtype", i32), ("causal_rows", i32)]


class ConcatArgs(C.Structure):
    _fields_ = [("a", vp), ("b", vp), ("y", vp), ("rows", i64), ("c1", i32), ("c2", i32), ("dtype", i32)]


class SiluArgs(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("n", i64)]


class CastArgs(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("rows", i64), ("cols", i32), ("ld", i32), ("dtype", i32)]


class UnetInputArgs(C.Structure):
    _fields_ = [("latents", vp), ("mask", vp), ("first", vp), ("x", vp),
                ("B", i32), ("F", i32), ("HW", i32), ("c_latent", i32), ("c_pad", i32), ("cfg_dup", i32), ("mask_frames", i32),
                ("dtype", i32), ("mode", i32)]


class CfgDdimArgs(C.Structure):
    _fields_ = [("pred", vp), ("latents", vp), ("coef", vp),
                ("B", i32), ("F", i32), ("HW", i32), ("c_latent", i32), ("ld", i32), ("cfg", i32), ("guidance", f32),
                ("pred_type", i32), ("clip_sample", i32), ("dtype", i32), ("pred_single", vp), ("video_scale", f32),
                ("variance_noise", vp), ("sigma", f32), ("clipped_model_output", i32)]


class NchwInArgs(C.Structure):
    _fields_ = [("z", vp), ("x", vp), ("N", i32), ("C", i32), ("HW", i32), ("c_pad", i32), ("scale", f32), ("dtype", i32)]


class NhwcOutArgs(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("N", i32), ("C", i32), ("HW", i32), ("ld", i32),
                ("mul", f32), ("add", f32), ("lo", f32), ("hi", f32), ("dtype", i32)]


class EmbedArgs(C.Structure):
    _fields_ = [("ids", vp), ("table", vp), ("pos", vp), ("out", vp), ("rows", i64), ("seq", i32), ("C", i32), ("vocab", i32),
                ("dtype", i32)]


class PatchifyArgs(C.Structure):
    _fields_ = [("image", vp), ("out", vp), ("B", i32), ("Cin", i32), ("H", i32), ("W", i32), ("P", i32), ("ld", i32), ("dtype", i32)]


class TemporalBlockArgs(C.Structure):
    _fields_ = [("x", vp), ("out", vp), ("w_qkv", vp), ("colsum", vp), ("bias", vp), ("pe_bias", vp), ("w_out", vp), ("b_out", vp),
                ("clips", i32), ("frames", i32), ("pixels", i32), ("heads", i32), ("d", i32), ("C", i32),
                ("scale", C.c_float), ("eps", C.c_float), ("dtype", i32), ("wstream", vp)]


class FFBlockArgs(C.Structure):
    _fields_ = [("x", vp), ("residual", vp), ("out", vp), ("wstream", vp), ("b_out", vp), ("chan_parts", vp), ("cs_rows", i32),
                ("rows", i32), ("C", i32), ("hidden", i32), ("eps", C.c_float), ("dtype", i32)]


class PanelLinearArgs(C.Structure):
    _fields_ = [("x", vp), ("residual", vp), ("out", vp), ("wstream", vp), ("bias", vp), ("gn_cs", vp), ("gn_gamma", vp), ("gn_beta", vp),
                ("gn_rows_per_sample", i32), ("gn_stat_samples", i32), ("gn_groups", i32), ("gn_eps", C.c_float),
                ("rows", i32), ("N", i32), ("K", i32), ("dtype", i32)]


class PackConv3x3Args(C.Structure):
    _fields_ = [("w", vp), ("out", vp), ("O", i32), ("I", i32), ("dtype", i32)]


class PackGegluArgs(C.Structure):
    _fields_ = [("w", vp), ("b", vp), ("w_out", vp), ("b_out", vp), ("O", i32), ("I", i32), ("dtype", i32)]


# name -> args struct for every `int fyc_<op>(const args*, void* stream)` entry point
OPS = {
    "fyc_gemm": GemmArgs, "fyc_attention": AttnArgs, "fyc_temporal_attention": TAttnArgs,
    "fyc_gn_stats": GnStatsArgs, "fyc_gn_apply": GnApplyArgs, "fyc_layernorm": LayerNormArgs,
    "fyc_softmax_rows": SoftmaxArgs, "fyc_concat_channels": ConcatArgs, "fyc_silu_f32": SiluArgs,
    "fyc_cast_from_f32": CastArgs, "fyc_cast_to_f32": CastArgs, "fyc_unet_input": UnetInputArgs,
    "fyc_cfg_ddim_step": CfgDdimArgs, "fyc_nchw_to_nhwc": NchwInArgs, "fyc_nhwc_to_nchw": NhwcOutArgs,
    "fyc_embed_tokens": EmbedArgs, "fyc_patchify": PatchifyArgs, "fyc_row_stats": RowStatsArgs,
    "fyc_gn_apply_cs": GnApplyCsArgs, "fyc_chan_stats_reduce": ChanStatsReduceArgs,
    "fyc_pack_conv3x3": PackConv3x3Args, "fyc_pack_geglu": PackGegluArgs, "fyc_temporal_block": TemporalBlockArgs,
    "fyc_ff_block": FFBlockArgs, "fyc_panel_linear": PanelLinearArgs,
}
MISC = ["fyc_version", "fyc_last_error", "fyc_init", "fyc_device_caps", "fyc_set_tuning", "fyc_gemm_row_parts", "fyc_gemm_stat_layout", "fyc_gemm_workspace_bytes", "fyc_gn_stats_workspace", "fyc_temporal_block_supported", "fyc_temporal_block_wstream_bytes",
        "fyc_ff_block_supported", "fyc_ff_block_wstream_bytes", "fyc_panel_linear_supported", "fyc_panel_linear_wstream_bytes"]

_lib = None
FYC_VERSION = 301        # the ABI version this binding's ctypes structs mirror (include/fyc.h::FYC_VERSION)


class FycError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the in-tree library; raise (never fall back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FycError(f"{LIB_PATH} is missing: build it with `python -m followyourclick_amd._build` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.fyc_version.restype = C.c_int
    ab_build = bool(os.environ.get("FYC_LIB_PATH"))      # an older library for A/B timing (tools/): newer entry points may be absent
    # (the whole number, not only the major: this binding mirrors exactly one header)
    if lib.fyc_version() != FYC_VERSION and not ab_build:
        raise FycError(f"{LIB_PATH} reports ABI version {lib.fyc_version()}, this binding was written against {FYC_VERSION}: argument structs differ "
                       "between major versions - rebuild the library (`python -m followyourclick_amd._build`)")
    lib.fyc_last_error.restype = C.c_char_p
    lib.fyc_init.argtypes = [vp]
    lib.fyc_device_caps.argtypes = [C.POINTER(i64)]
    lib.fyc_set_tuning.argtypes = [C.c_int, C.c_int]
    if not ab_build or hasattr(lib, "fyc_gemm_row_parts"):
        lib.fyc_gemm_row_parts.argtypes = [C.POINTER(GemmArgs)]
        lib.fyc_gemm_row_parts.restype = C.c_int
        lib.fyc_gemm_stat_layout.argtypes = [C.POINTER(GemmArgs), C.POINTER(i32), C.POINTER(i32)]
        lib.fyc_gemm_stat_layout.restype = C.c_int
        lib.fyc_gemm_workspace_bytes.argtypes = [C.POINTER(GemmArgs)]
        lib.fyc_gemm_workspace_bytes.restype = i64
    if not ab_build or hasattr(lib, "fyc_gn_stats_workspace"):
        lib.fyc_gn_stats_workspace.argtypes = [C.POINTER(GnStatsArgs)]
        lib.fyc_gn_stats_workspace.restype = i64
    if not ab_build or hasattr(lib, "fyc_temporal_block_supported"):
        lib.fyc_temporal_block_supported.argtypes = [C.POINTER(TemporalBlockArgs)]
        lib.fyc_temporal_block_supported.restype = C.c_int
    if not ab_build or hasattr(lib, "fyc_temporal_block_wstream_bytes"):
        lib.fyc_temporal_block_wstream_bytes.argtypes = []
        lib.fyc_temporal_block_wstream_bytes.restype = i64
    if not ab_build or hasattr(lib, "fyc_ff_block_supported"):
        lib.fyc_ff_block_supported.argtypes = [C.POINTER(FFBlockArgs)]
        lib.fyc_ff_block_supported.restype = C.c_int
        lib.fyc_ff_block_wstream_bytes.argtypes = []
        lib.fyc_ff_block_wstream_bytes.restype = i64
    if not ab_build or hasattr(lib, "fyc_panel_linear_supported"):
        lib.fyc_panel_linear_supported.argtypes = [C.POINTER(PanelLinearArgs)]
        lib.fyc_panel_linear_supported.restype = C.c_int
        lib.fyc_panel_linear_wstream_bytes.argtypes = [i32, i32]
        lib.fyc_panel_linear_wstream_bytes.restype = i64
    for name, st in OPS.items():
        if ab_build and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.argtypes = [C.POINTER(st), vp]
        fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise FycError(f"{what} failed (rc={rc}): {load().fyc_last_error().decode(errors='replace')}")
