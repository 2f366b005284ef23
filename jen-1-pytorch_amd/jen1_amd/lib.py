"""ctypes binding of libjen1_hip.so (C ABI declared in include/jen1_hip.h).

There is NO fallback: if the shared library has not been built (``python -c
"import __graft_entry__ as g; g.build()"``) every product entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(HERE)
REPO_ROOT = os.path.dirname(PKG_ROOT)
ABI_VERSION = 2          # layout of the structs mirrored below (GemmArgs, RepackEntry, ...): bumped together with jen1_abi_version()
LIB_PATH = os.environ.get("JEN1_LIB", os.path.join(HERE, "libjen1_hip.so"))   # JEN1_LIB: tuning builds only
CSRC = os.path.join(PKG_ROOT, "csrc")
INCLUDE = os.path.join(REPO_ROOT, "include")
SOURCES = ["conv_gemm.hip", "stream_gemm.hip", "tile_gemm.hip", "norm_apply.hip", "attention.hip", "deep_kernel.hip", "elementwise.hip",
           "optimizer.hip", "train_gemm.hip", "train_ops.hip", "train_attn.hip", "encodec.hip", "big_gemm.hip", "train_glue.hip", "train_kvbank.hip",
           "long_kernel.hip"]

F32, BF16, FP8 = 0, 1, 2
PRO_NONE, PRO_GN, PRO_GN_SILU, PRO_LN, PRO_SILU = 0, 1, 2, 3, 4
ACT_NONE, ACT_GELU = 0, 1
CFG_W64x64, CFG_W128x64, CFG_S16x64, CFG_S16x32, CFG_S16x16 = 0, 1, 2, 3, 4
CFG_T128x64, CFG_T128x32, CFG_T128x16, CFG_T256x32, CFG_T256x16, CFG_T64x64 = 5, 6, 7, 8, 9, 10

c_void_p, c_int, c_float, c_int64 = C.c_void_p, C.c_int32, C.c_float, C.c_int64


MAX_SEG = 16


class ConvSeg(C.Structure):
    """mirror of ``jen1_conv_seg``."""
    _fields_ = [("x", c_void_p), ("ld", c_int), ("shift", c_int), ("kch", c_int), ("reserved", c_int)]


class ConvArgs(C.Structure):
    """mirror of ``jen1_conv_args`` (include/jen1_hip.h) -- field order is ABI."""
    _fields_ = [
        ("x0", c_void_p), ("x1", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("residual", c_void_p),
        ("y", c_void_p), ("gn_stats0", c_void_p), ("gn_stats1", c_void_p), ("gn_gamma", c_void_p),
        ("gn_beta", c_void_p), ("film", c_void_p), ("film_row", c_void_p), ("ln_rowstats", c_void_p),
        ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("row_scale", c_void_p), ("out_gn_stats", c_void_p),
        ("out_rowstats", c_void_p), ("slab", c_void_p), ("counters", c_void_p),
        ("dtype", c_int), ("B", c_int), ("L_in", c_int), ("L_out", c_int),
        ("c0", c_int), ("c1", c_int), ("ld0", c_int), ("ld1", c_int),
        ("taps", c_int), ("stride", c_int), ("pad_left", c_int),
        ("M", c_int), ("out_C", c_int), ("ps_f", c_int), ("ps_off", c_int),
        ("L_y", c_int), ("y_brows", c_int), ("y_row0", c_int), ("ld_y", c_int), ("ld_res", c_int),
        ("y_f32", c_int), ("pro_mode", c_int),
        ("gn_groups", c_int), ("gn_cpg", c_int), ("gn_count", c_int),
        ("gn_eps", c_float), ("src1_scale", c_float),
        ("film_off", c_int), ("film_C", c_int), ("film_ld", c_int),
        ("ln_C", c_int), ("ln_eps", c_float),
        ("act", c_int), ("out_cpf", c_int), ("tb", c_int), ("nb", c_int),
        ("kc_stage", c_int), ("splitk", c_int), ("cfg", c_int), ("direct", c_int),
        ("zeros", c_void_p), ("tiles_t", c_int), ("inv_tiles_t", c_float), ("inv_tb", c_float),
        ("film_step", c_void_p), ("ln_u", c_void_p), ("ln_fold", c_int), ("nseg", c_int),
        ("seg", ConvSeg * MAX_SEG), ("m_split", c_int), ("k_split", c_int), ("w_scale", c_void_p),
        ("live_mask", c_int), ("reserved_", c_int),
    ]


class NormArgs(C.Structure):
    """mirror of ``jen1_norm_args`` (include/jen1_hip.h)."""
    _fields_ = [
        ("x0", c_void_p), ("x1", c_void_p), ("y", c_void_p), ("gn_stats0", c_void_p), ("gn_stats1", c_void_p),
        ("gamma", c_void_p), ("beta", c_void_p), ("film", c_void_p), ("film_row", c_void_p), ("ln_rowstats", c_void_p),
        ("dtype", c_int), ("mode", c_int), ("B", c_int), ("L", c_int), ("c0", c_int), ("c1", c_int),
        ("ld0", c_int), ("ld1", c_int), ("ld_y", c_int), ("groups", c_int), ("cpg", c_int), ("count", c_int),
        ("eps", c_float), ("src1_scale", c_float), ("film_off", c_int), ("film_C", c_int), ("film_ld", c_int),
        ("film_step", c_void_p),
    ]


class GemmOperand(C.Structure):
    """mirror of ``jen1_gemm_operand`` (include/jen1_train.h)."""
    _fields_ = [("p", c_void_p), ("zs0", c_int64), ("zs1", c_int64), ("ld_r", c_int64), ("ld_k", c_int64),
                ("tap_stride", c_int64), ("zdiv", c_int), ("map_axis", c_int), ("map_L", c_int), ("map_Lsrc", c_int),
                ("map_mul", c_int), ("map_tapmul", c_int), ("map_shift", c_int), ("map_div", c_int),
                ("map_reflect", c_int), ("reserved", c_int)]


class GemmArgs(C.Structure):
    """mirror of ``jen1_gemm_args`` (include/jen1_train.h)."""
    _fields_ = [("a", GemmOperand), ("b", GemmOperand), ("c", c_void_p), ("bias", c_void_p),
                ("c_zs0", c_int64), ("c_zs1", c_int64), ("ldc_m", c_int64), ("ldc_n", c_int64), ("c_tap_stride", c_int64),
                ("c_zdiv", c_int), ("M", c_int), ("N", c_int), ("K", c_int), ("taps", c_int), ("batches", c_int),
                ("taps_in_z", c_int), ("splitk", c_int), ("atomic", c_int), ("accumulate", c_int), ("c_f32", c_int),
                ("dtype", c_int), ("alpha", c_float), ("reserved", c_int), ("rowsum", c_void_p), ("residual", c_void_p),
                ("map_shift_b", c_void_p)]


class BGemmGroup(C.Structure):
    """mirror of ``jen1_bgemm_group`` (include/jen1_hip.h): one column group = one output tensor of the grouped GEMM"""
    _fields_ = [("c", c_void_p), ("bias", c_void_p), ("n0", c_int), ("N", c_int), ("ldc", c_int), ("reserved", c_int)]


class BGemmArgs(C.Structure):
    """mirror of ``jen1_bgemm_args`` (include/jen1_hip.h)"""
    _fields_ = [("a", c_void_p), ("b", c_void_p), ("groups", c_void_p), ("row_scale", c_void_p),
                ("M", c_int), ("Ntot", c_int), ("K", c_int), ("lda", c_int), ("ldb", c_int), ("n_groups", c_int),
                ("rows_in", c_int), ("rows_out", c_int), ("c_f32", c_int), ("accumulate", c_int), ("dtype", c_int), ("ldc", c_int),
                ("alpha", c_float), ("group_align", c_int), ("c", c_void_p), ("bias", c_void_p)]


class RepackEntry(C.Structure):
    """mirror of ``jen1_repack_entry`` (include/jen1_train.h)."""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("d0", c_int), ("d1", c_int), ("d2", c_int), ("ld", c_int),
                ("s0", c_int64), ("s1", c_int64), ("s2", c_int64), ("tile0", c_int), ("ld2", c_int), ("dst2", c_void_p)]


class KvLayer(C.Structure):
    """mirror of ``jen1_kv_layer`` (include/jen1_train.h)"""
    _fields_ = [("w", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("gw", c_void_p), ("ggamma", c_void_p), ("gbeta", c_void_p),
                ("n0", c_int), ("N", c_int)]


# every symbol include/jen1_hip.h and include/jen1_train.h declare: (name, restype, argtypes)
_P = c_void_p
SYMBOLS = {
    "jen1_conv_gemm": (c_int, [C.POINTER(ConvArgs), _P]),
    "jen1_conv_gemm_lds_bytes": (c_int64, [C.POINTER(ConvArgs)]),
    "jen1_norm_apply": (c_int, [C.POINTER(NormArgs), _P]),
    "jen1_cfg_bm": (c_int, [c_int]),
    "jen1_cfg_bn": (c_int, [c_int]),
    "jen1_attention": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int] + [c_int] * 12 + [c_float, c_int, _P]),
    "jen1_attention_fin": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int] + [c_int] * 12 + [c_float, _P, _P, _P, c_int, c_float, c_int, c_int, c_int, _P]),
    "jen1_pack_input": (c_int, [_P, _P, _P, _P] + [c_int] * 7 + [_P]),
    "jen1_pack_input_parts": (c_int, [_P, _P, _P, _P] + [c_int] * 7 + [_P]),
    "jen1_gn_stats_from_parts": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "jen1_unpack_output": (c_int, [_P, _P] + [c_int] * 5 + [_P]),
    "jen1_row_stats": (c_int, [_P, _P] + [c_int] * 4 + [_P]),
    "jen1_gn_stats": (c_int, [_P, _P] + [c_int] * 4 + [_P]),
    "jen1_time_features": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "jen1_time_features_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "jen1_linear_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "jen1_cfg_ddim_step": (c_int, [_P] * 8 + [c_int] * 5 + [c_float, c_int, c_float, c_int, c_int, c_int, _P]),
    "jen1_step_advance": (c_int, [_P, _P]),
    "jen1_cfg_ddim_step_adv": (c_int, [_P] * 9 + [c_int] * 5 + [c_float, c_int, c_float, c_int, c_int, c_int, _P]),
    "jen1_cfg_ddim_step_pack": (c_int, [_P] * 9 + [c_int] * 6 + [c_float, c_int, c_float, c_int, c_int, c_int, _P]),
    "jen1_step_tail": (c_int, [_P] * 9 + [c_int] * 6 + [c_float, c_int, c_float, c_int, c_int, c_int, _P, c_int, _P, _P, c_int64, _P]),
    "jen1_cfg_combine": (c_int, [_P, _P] + [c_int] * 4 + [c_float, c_int, c_float, c_int, _P]),
    "jen1_grad_sqnorm": (c_int, [_P, c_int64, _P, _P]),
    "jen1_grad_sqnorm_scratch_bytes": (c_int64, []),
    "jen1_grad_sqnorm_ws": (c_int, [_P, c_int64, _P, _P, _P]),
    "jen1_adamw_step": (c_int, [_P, _P, _P, _P, c_int64] + [c_float] * 5 + [c_int, _P, c_float, c_int, _P]),
    "jen1_adamw_step_counted": (c_int, [_P, _P, _P, _P, c_int64] + [c_float] * 5 + [_P, _P, c_float, c_int, _P]),
    "jen1_memset_zero": (c_int, [_P, c_int64, _P]),
    "jen1_big_gemm": (c_int, [C.POINTER(BGemmArgs), _P]),
    "jen1_big_gemm_tn": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "jen1_big_gemm_tn_store": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "jen1_big_gemm_conv": (c_int, [_P] * 5 + [c_int] * 13 + [_P, c_int, _P]),
    "jen1_big_gemm_tn_conv": (c_int, [_P, _P, _P, _P] + [c_int] * 10 + [c_float, _P, _P]),
    "jen1_standardize_rows": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    "jen1_kv_fixed_fill": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P]),
    "jen1_train_gemm": (c_int, [C.POINTER(GemmArgs), _P]),
    "jen1_train_gemm_pair": (c_int, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), _P]),
    "jen1_attn_small_fits": (c_int, [c_int, c_int, c_int, c_int]),
    "jen1_attn_small_forward": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64] + [c_int] * 5 + [c_float, c_int, _P, _P, c_int, _P]),
    "jen1_attn_small_backward": (c_int, [_P, c_int64] * 8 + [c_int] * 5 + [c_float, _P, c_int, _P]),
    "jen1_attn_small_forward_rows": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64] + [c_int] * 5 + [c_float, c_int, _P, _P, _P, c_int, _P]),
    "jen1_attn_small_backward_rows": (c_int, [_P, c_int64] * 8 + [c_int] * 5 + [c_float, _P, _P, c_int, _P]),
    "jen1_sum_rows_inplace": (c_int, [_P, c_int, c_int64, c_int, _P]),
    "jen1_gn_sums": (c_int, [_P, _P] + [c_int] * 6 + [_P]),
    "jen1_gn_apply": (c_int, [_P, _P, _P, _P, _P, c_int, _P] + [c_int] * 5 + [c_float, c_int, c_int, _P]),
    "jen1_gn_backward": (c_int, [_P] * 6 + [c_int] + [_P] * 6 + [c_int] * 5 + [c_float, c_int, c_int, _P]),
    "jen1_gn_backward_add": (c_int, [_P] * 6 + [c_int] + [_P] * 7 + [c_int] * 5 + [c_float, c_int, c_int, _P]),
    "jen1_gn_backward_add2": (c_int, [_P] * 6 + [c_int] + [_P] * 8 + [c_int] * 5 + [c_float, c_int, c_int, _P]),
    "jen1_ln_backward_add": (c_int, [_P] * 8 + [c_int] * 4 + [_P]),
    "jen1_ln2_forward": (c_int, [_P] * 8 + [c_int] * 3 + [C.c_float, c_int, _P]),
    "jen1_ln2_backward_add": (c_int, [_P] * 12 + [c_int] * 4 + [_P]),
    "jen1_repack": (c_int, [_P, c_int, c_int, c_int, _P]),
    "jen1_kv_fold": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, _P, _P]),
    "jen1_kv_fold_backward": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "jen1_sum_rows_strided": (c_int, [_P, c_int, c_int, c_int, c_int64, c_int, _P]),
    "jen1_train_pack_input": (c_int, [_P] * 6 + [c_int] * 6 + [_P, _P, _P, c_int, _P]),
    "jen1_train_context": (c_int, [_P] * 5 + [c_int] * 6 + [_P]),
    "jen1_train_context_backward": (c_int, [_P] * 4 + [c_int] * 6 + [_P]),
    "jen1_time_features_fwd": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, _P]),
    "jen1_time_features_bwd": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, c_int, _P]),
    "jen1_cfg_loss_forward": (c_int, [_P, _P, _P] + [c_int] * 5 + [c_float, c_int, c_float, c_int, c_int, _P]),
    "jen1_cfg_loss_backward": (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [c_float, c_int, c_float, c_int, c_int, _P]),
    "jen1_concat2": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_float, c_int, _P]),
    "jen1_split2": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_float, c_int, _P]),
    "jen1_gn_forward": (c_int, [_P, _P, _P, _P, _P, c_int, _P] + [c_int] * 5 + [c_float, c_int, c_int, _P]),
    "jen1_ln_forward": (c_int, [_P] * 5 + [c_int] * 3 + [c_float, c_int, _P]),
    "jen1_ln_backward": (c_int, [_P] * 7 + [c_int] * 4 + [_P]),
    "jen1_act_forward": (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    "jen1_act_backward": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P]),
    "jen1_softmax_forward": (c_int, [_P, _P] + [c_int] * 7 + [_P]),
    "jen1_softmax_backward": (c_int, [_P, _P, _P] + [c_int] * 5 + [_P]),
    "jen1_colsum": (c_int, [_P, _P] + [c_int] * 4 + [_P]),
    "jen1_convert_clear": (c_int, [_P, _P, c_int64, c_int, _P]),
    "jen1_convert_clear_add": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "jen1_rvq_decode": (c_int, [_P, _P, _P] + [c_int] * 5 + [_P]),
    "jen1_lstm_layer": (c_int, [_P, _P, _P, _P] + [c_int] * 5 + [_P]),
    "jen1_lstm_layer_multi": (c_int, [_P, _P, _P, _P, _P, _P] + [c_int] * 5 + [_P]),
    # include/jen1_deep.h: the persistent deep-level kernel (phase descriptors are opaque bytes on this side)
    "jen1_deep_phase_size": (c_int, []),
    "jen1_deep_has_chunks": (c_int, []),
    "jen1_deep_phase_conv": (c_int, [C.POINTER(ConvArgs), c_int, _P]),
    "jen1_deep_phase_attention": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int] + [c_int] * 12 +
                                  [c_float, _P, _P, c_int, c_float, c_int, c_int, c_int, c_int, _P]),
    "jen1_deep_phase_stats": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "jen1_deep_phase_tile": (c_int, [C.POINTER(ConvArgs), c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, _P, c_int, _P]),
    "jen1_deep_tile_count": (c_int, [c_int, c_int, c_int, c_int]),
    "jen1_deep_link": (c_int, [_P, c_int, c_int, _P, _P]),
    "jen1_deep_poison": (c_int, [_P, c_int, _P, _P]),
    "jen1_deep_poison_zero": (c_int, [_P, c_int, _P, _P, c_int64, _P]),
    "jen1_deep_blob_bytes": (c_int, []),
    "jen1_deep_sync_bytes": (c_int64, [c_int]),
    "jen1_deep_num_workgroups": (c_int, []),
    "jen1_deep_run": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, _P]),
    "jen1_deep_run_err": (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, c_int, _P]),
    "jen1_deep_error_word": (c_int, [c_int]),
    "jen1_deep_run_mode": (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "jen1_deep_run_kinds": (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    # include/jen1_long.h: the sample-resident long-level kernel
    "jen1_long_geometry": (c_int, [c_int, c_int, c_int, C.POINTER(c_int), C.POINTER(c_int), C.POINTER(c_int)]),
    "jen1_long_phase_conv": (c_int, [C.POINTER(ConvArgs), c_int, _P, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "jen1_long_phase_lds": (c_int, [_P]),
    "jen1_long_phase_units": (c_int, [_P]),
    "jen1_long_run": (c_int, [_P, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "jen1_long_census": (c_int, [_P, c_int, _P]),
    "jen1_long_debug_buffer": (c_int, [_P]),
    "jen1_last_error": (C.c_char_p, []),
    "jen1_build_info": (C.c_char_p, []),
    "jen1_abi_version": (c_int, []),
}

_lib: Optional[C.CDLL] = None


class Jen1HipError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into jen1_amd/libjen1_hip.so (hipcc cross-compiles without a GPU): one object per source
    under <package>/build/ (compiled in parallel, only when the source or a header is newer than its object), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(INCLUDE, "jen1_hip.h"), os.path.join(INCLUDE, "jen1_train.h"),
            os.path.join(INCLUDE, "jen1_deep.h"), os.path.join(INCLUDE, "jen1_long.h")]
    hdrs += [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h") and h != "common.h"]
    deps = srcs + hdrs
    if os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}"] + os.environ.get("JEN1_HIPCC_FLAGS", "").split()
    objdir = os.path.join(PKG_ROOT, "build", "obj" + ("" if not os.environ.get("JEN1_HIPCC_FLAGS") else "_" + str(abs(hash(os.environ["JEN1_HIPCC_FLAGS"])) % 10 ** 8)))
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in hdrs)

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time):
            return obj
        cmd = [hipcc, *flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise Jen1HipError(f"hipcc failed on {os.path.basename(src)}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=int(os.environ.get("JEN1_BUILD_JOBS", "6"))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise Jen1HipError(f"hipcc link failed:\n{r.stdout}\n{r.stderr}")
    global _lib
    _lib = None
    return LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Jen1HipError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` from the repo root. "
            "There is no CPU / eager fallback for the denoiser path.")
    # torch first: its bundled libamdhip64 must be the HIP runtime of the process.  Loading libjen1_hip.so before torch
    # pulls in the system runtime instead, and two runtimes in one process fail with "no ROCm-capable device".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # raises AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.jen1_abi_version() != ABI_VERSION:
        raise Jen1HipError("libjen1_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().jen1_last_error()
        raise Jen1HipError(f"{what}: {msg.decode() if msg else 'unknown error'}")


def bgemm_group_table(groups, device):
    """device copy of a ``jen1_bgemm_group`` table: ``groups`` = [(c_ptr, bias_ptr or None, n0, N, ldc)]"""
    import numpy as np
    import torch
    arr = (BGemmGroup * len(groups))()
    for i, (c, bias, n0, N, ldc) in enumerate(groups):
        arr[i].c, arr[i].bias, arr[i].n0, arr[i].N, arr[i].ldc = c, bias, n0, N, ldc
    return torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(device)
