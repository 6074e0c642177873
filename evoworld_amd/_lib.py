"""ctypes binding of libevoworld_hip.so (the C ABI in include/evoworld_hip.h).

The library is built in-tree by `make -C evoworld_amd/csrc` (or __graft_entry__.build()).  Loading fails
loudly -- there is no CPU / PyTorch fallback for the product path.
"""
import ctypes
import os

import torch  # noqa: F401  MUST precede the CDLL below: torch bundles its own libamdhip64.so.7; loading ours first
#                     would pull /opt/rocm's copy and leave two HIP runtimes in the process ("no ROCm-capable device").
from ctypes import c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EW_LIB_PATH") or os.path.join(_HERE, "libevoworld_hip.so")   # EW_LIB_PATH: another build of the same ABI (A/B tools)
ABI_VERSION = 10

# every symbol declared in include/evoworld_hip.h
SYMBOLS = [
    "ew_abi_version", "ew_last_error", "ew_gemm_f16", "ew_set_gemm_generation", "ew_get_gemm_generation", "ew_set_gemm_debug", "ew_gemm_last_kernel", "ew_gemm_streamk_status", "ew_gemm_streamk_init", "ew_ff_geglu320_f16", "ew_groupnorm_workspace_floats", "ew_groupnorm_stats_f16", "ew_groupnorm_finalize", "ew_groupnorm_apply_f16",
    "ew_layernorm_f16", "ew_attn_spatial_f16", "ew_attn_spatial_log2_f16", "ew_attn_temporal_f16", "ew_nchw_f32_to_nhwc_f16",
    "ew_nhwc_f16_to_nchw_f32", "ew_softmax_rows_f16", "ew_time_conv3_f32", "ew_euler_cfg_step", "ew_plucker_embed", "ew_cube2equi_gather",
    "ew_depth_unproject", "ew_select_workspace_bytes", "ew_select_kth_f32", "ew_filter_compact_workspace_bytes",
    "ew_filter_compact", "ew_splat_cubemap", "ew_splat_resolve", "ew_equi2pers", "ew_resize_aa_u8",
    "ew_u8_hwc_to_f32_chw", "ew_f32_chw_to_u8_hwc", "ew_blur_axis_f32", "ew_bicubic_resize_f32", "ew_vit_patchify_f16",
    "ew_attn_small_f16",
    "ew_set_cu_budget", "ew_get_cu_budget", "ew_stream_create_cu_mask", "ew_stream_destroy",
    "ew_nchw_f32_to_nhwc_split_f16", "ew_euler_cfg_step_split", "ew_groupnorm_apply_split_f16", "ew_sinusoid_embed_f16",
]


class GemmArgs(ctypes.Structure):
    """struct ew_gemm_args (include/evoworld_hip.h)."""
    _fields_ = [
        ("a", c_void_p), ("a2", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("rowbias", c_void_p),
        ("r1", c_void_p), ("r2", c_void_p), ("out", c_void_p), ("zero_page", c_void_p),
        ("M", c_int), ("N", c_int), ("c1", c_int), ("c2", c_int), ("lda", c_int), ("lda2", c_int),
        ("ld_out", c_int), ("ld_r1", c_int), ("ld_r2", c_int), ("ld_rowbias", c_int), ("mode", c_int),
        ("n_img", c_int), ("h_in", c_int), ("w_in", c_int), ("h_out", c_int), ("w_out", c_int),
        ("stride", c_int), ("upsample", c_int), ("tB", c_int), ("tT", c_int), ("tP", c_int),
        ("rows_per_group", c_int), ("act", c_int), ("c_acc", c_float), ("c_r1", c_float), ("c_r2", c_float),
        ("r1_lo", c_void_p), ("r2_lo", c_void_p), ("out_lo", c_void_p), ("conv_shift", c_int),
    ]


class FfArgs(ctypes.Structure):
    """struct ew_ff_args (include/evoworld_hip.h)."""
    _fields_ = [
        ("x", c_void_p), ("w1p", c_void_p), ("b1p", c_void_p), ("w2p", c_void_p), ("b2", c_void_p), ("rowbias", c_void_p),
        ("r1", c_void_p), ("r1_lo", c_void_p), ("r2", c_void_p), ("r2_lo", c_void_p), ("out", c_void_p), ("out_lo", c_void_p),
        ("zero_page", c_void_p), ("M", c_int), ("C", c_int), ("hidden", c_int), ("rows_per_group", c_int), ("ld_rowbias", c_int),
        ("c_acc", c_float), ("c_r1", c_float), ("c_r2", c_float),
    ]


_lib = None


class EvoWorldHipError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise (never fall back) if it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EvoWorldHipError(
            f"{LIB_PATH} not found: build it with `make -C evoworld_amd/csrc` (hipcc --offload-arch=gfx950). "
            "evoworld_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for s in SYMBOLS:       # (EW_LIB_PATH: another BUILD of the same ABI -- A/B of two compilations in one gpurun call; older ABIs are refused)
        if not hasattr(lib, s):
            raise EvoWorldHipError(f"{LIB_PATH} does not export {s}")
    lib.ew_last_error.restype = c_char_p
    lib.ew_gemm_last_kernel.restype = c_char_p
    lib.ew_abi_version.restype = c_int
    if lib.ew_abi_version() != ABI_VERSION:
        raise EvoWorldHipError(f"ABI mismatch: library {lib.ew_abi_version()} != binding {ABI_VERSION}")
    P, I, F, LL = c_void_p, c_int, c_float, c_longlong
    sig = {
        "ew_gemm_f16": [ctypes.POINTER(GemmArgs), P],
        "ew_groupnorm_stats_f16": [P, P, P, I, I, I, I, I, I, P],
        "ew_groupnorm_finalize": [P, I, I, I, I, P],
        "ew_groupnorm_apply_f16": [P, P, P, P, P, P, I, I, I, I, I, I, F, I, P],
        "ew_groupnorm_apply_split_f16": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, F, I, P],
        "ew_gemm_streamk_init": [P],
        "ew_ff_geglu320_f16": [ctypes.POINTER(FfArgs), P],
        "ew_layernorm_f16": [P, P, P, I, P, P, P, P, P, I, I, F, P],
        "ew_attn_spatial_f16": [P, P, P, P, I, I, I, I, LL, I, F, P],
        "ew_attn_spatial_log2_f16": [P, P, P, P, I, I, I, I, LL, I, P],
        "ew_attn_temporal_f16": [P, P, P, P, I, I, I, I, I, I, F, P],
        "ew_nchw_f32_to_nhwc_f16": [P, P, I, I, I, I, I, I, F, P],
        "ew_nhwc_f16_to_nchw_f32": [P, P, I, I, I, I, I, P],
        "ew_euler_cfg_step": [P, I, P, P, F, F, P, I, I, I, I, P],
        "ew_nchw_f32_to_nhwc_split_f16": [P, P, I, I, I, I, I, I, I, I, F, P],
        "ew_euler_cfg_step_split": [P, I, P, P, F, F, P, I, I, I, I, I, I, P],
        "ew_sinusoid_embed_f16": [P, I, I, I, P, P],
        "ew_softmax_rows_f16": [P, P, P, LL, I, LL, P],
        "ew_time_conv3_f32": [P, P, P, P, I, I, I, I, P],
        "ew_plucker_embed": [P, P, P, I, I, I, P],
        "ew_cube2equi_gather": [P, I, P, P, I, I, I, I, P],
        "ew_select_kth_f32": [P, c_size_t, c_size_t, P, P, P],
        "ew_filter_compact": [P, c_size_t, F, P, P, I, ctypes.c_uint, P, P, P, P, P],
        "ew_depth_unproject": [P, P, P, P, I, I, I, P],
        "ew_splat_cubemap": [P, c_size_t, P, P, I, I, F, F, F, F, F, P],
        "ew_splat_resolve": [P, P, I, P, I, I, I, P],
        "ew_equi2pers": [P, P, P, I, I, I, I, I, F, P],
        "ew_resize_aa_u8": [P, P, P, P, P, I, P, P, I, I, I, I, I, I, P],
        "ew_u8_hwc_to_f32_chw": [P, P, I, I, I, P],
        "ew_f32_chw_to_u8_hwc": [P, P, I, I, I, P],
        "ew_blur_axis_f32": [P, P, I, P, LL, I, I, I, P],
        "ew_bicubic_resize_f32": [P, P, I, I, I, I, I, I, P, P, P],
        "ew_vit_patchify_f16": [P, P, I, I, I, I, P],
        "ew_attn_small_f16": [P, P, P, P, I, I, I, I, I, I, F, P],
    }
    lib.ew_groupnorm_workspace_floats.argtypes = [c_int, c_int, c_int, c_int]
    lib.ew_groupnorm_workspace_floats.restype = c_size_t
    lib.ew_select_workspace_bytes.argtypes = []
    lib.ew_select_workspace_bytes.restype = c_size_t
    lib.ew_filter_compact_workspace_bytes.argtypes = [c_size_t]
    lib.ew_filter_compact_workspace_bytes.restype = c_size_t
    lib.ew_set_gemm_generation.argtypes = [c_int]
    lib.ew_set_gemm_generation.restype = None
    lib.ew_get_gemm_generation.restype = c_int
    lib.ew_gemm_streamk_status.argtypes = []
    lib.ew_gemm_streamk_status.restype = c_int
    lib.ew_set_gemm_debug.argtypes = [c_int]
    lib.ew_set_gemm_debug.restype = None
    lib.ew_set_cu_budget.argtypes, lib.ew_set_cu_budget.restype = [c_int], c_int
    lib.ew_get_cu_budget.argtypes, lib.ew_get_cu_budget.restype = [], c_int
    lib.ew_stream_create_cu_mask.argtypes, lib.ew_stream_create_cu_mask.restype = [c_int, c_int], c_void_p
    lib.ew_stream_destroy.argtypes, lib.ew_stream_destroy.restype = [c_void_p], c_int
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().ew_last_error()
        raise EvoWorldHipError(f"{what} failed ({status}): {msg.decode() if msg else ''}")
