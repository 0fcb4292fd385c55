"""ctypes binding of libdtlr_hip.so (include/dtlr_hip.h).  There is no fallback: if the shared
object is missing or a symbol is absent, importing/using the product path raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int, c_int64, c_void_p, c_float, POINTER

HERE = os.path.dirname(os.path.abspath(__file__))
# DTLR_HIP_LIB: profiling tools point this at the instrumented build (dtlr_amd/build.py --instr)
LIB_PATH = os.environ.get("DTLR_HIP_LIB") or os.path.join(HERE, "libdtlr_hip.so")
# the same sources compiled with IEEE fp16 as the library's 16-bit format (csrc/dtlr_common.h, DTLR_HALF_IS_F16): identical
# symbols, accepts DTLR_F16 wherever libdtlr_hip.so accepts DTLR_BF16
LIB_PATH_F16 = os.environ.get("DTLR_HIP_LIB_F16") or os.path.join(HERE, "libdtlr_hip_f16.so")

DTLR_F32, DTLR_F64, DTLR_BF16, DTLR_F16, DTLR_F32S = 0, 1, 2, 3, 4

_lib = None
_lib_f16 = None


class DTLRError(RuntimeError):
    pass


class K256sSlice(ctypes.Structure):
    """include/dtlr_hip.h: dtlr_k256s_slice"""
    _fields_ = [("Wp", c_void_p), ("bias", c_void_p), ("R", c_void_p), ("C", c_void_p),
                ("ldc", c_int), ("ldr", c_int), ("n_valid", c_int), ("relu", c_int)]


# name -> (restype, argtypes): every symbol include/dtlr_hip.h declares
_SIGNATURES = {
    "dtlr_strerror": (c_char_p, [c_int]),
    "dtlr_last_hip_error": (c_int, []),
    "dtlr_abi_version": (c_int, []),
    "dtlr_workspace_reserve": (c_int, [ctypes.c_long, c_void_p]),
    "dtlr_workspace_retired_bytes": (ctypes.c_long, []),
    "dtlr_msda_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dtlr_msda_fused_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                        c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dtlr_msda_fused_forward_strided": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dtlr_msda_encoder_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_int, c_void_p, c_void_p]),
    "dtlr_msda_encoder_plan_ok": (c_int, [c_void_p, c_int, c_int]),
    "dtlr_msda_encoder_far_samples": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dtlr_swin_patch_embed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dtlr_swin_window_attn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_swin_patch_merge": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dtlr_geometry": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dtlr_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_long, c_int, c_float, c_int, c_void_p]),
    "dtlr_ffn_fused_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                    c_int, c_int, c_int, c_void_p]),
    "dtlr_conv3x3_patch_supported": (c_int, [c_int, c_int]),
    "dtlr_conv3x3_patch_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_conv3x3_patch_f32s_supported": (c_int, [c_int, c_int]),
    "dtlr_conv3x3_patch_f32s": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_gemm_kres_pack_weights": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "dtlr_gemm_kres_pack_weights_bcast384": (c_int, [c_void_p, c_void_p]),
    "dtlr_gemm_kres_bcast384": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "dtlr_gemm_kres": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_gemm_kres_chain": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dtlr_gemm_kres_cat_s2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_ffn32_pack_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "dtlr_ffn32_pad_chunks": (c_int, []),
    "dtlr_ffn32_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                ctypes.c_long, c_int, c_void_p]),
    "dtlr_ffn4_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                               ctypes.c_long, c_int, c_void_p]),
    "dtlr_ffn_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, ctypes.c_long, c_int, c_void_p]),
    "dtlr_ffn_split_pad_chunks": (c_int, []),
    "dtlr_head_ts": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, ctypes.c_long, c_void_p]),
    "dtlr_head_ts_pad_chunks": (c_int, []),
    "dtlr_k256s_pack_weights": (c_int, [c_void_p, c_void_p, c_void_p]),
    "dtlr_gemm_k256s": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, ctypes.c_long, c_void_p]),
    "dtlr_gemm_k256s_multi": (c_int, [c_void_p, ctypes.c_long, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "dtlr_proj_pack_weights": (c_int, [c_void_p, c_void_p]),
    "dtlr_proj_ln_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p]),
    "dtlr_proj_ln_split_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p]),
    "dtlr_mha_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_mha_workspace_bytes": (ctypes.c_long, [c_int, c_int, c_int, c_int]),
    "dtlr_gemm_nt": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_split_pack_weights": (c_int, [c_void_p, c_void_p, ctypes.c_long, c_int, c_void_p]),
    "dtlr_k256_pack_weights": (c_int, [c_void_p, c_void_p, c_int]),
    "dtlr_gemm_k256": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dtlr_proj_ln_k256_pack_weights": (c_int, [c_void_p, c_void_p]),
    "dtlr_proj_ln_k256": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p]),
    "dtlr_gemm_nt_a2bcast": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_gemm_nt_resbcast": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_gemm_nt_rowmax": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_gemm_nt_rowmax_lda": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_two_stage_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_conv2d_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_dq_pack_weights": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "dtlr_dec_query_stage": (c_int, [c_void_p] * 16 + [c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_decoder_query_prep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_box_refine": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_long, c_void_p]),
    "dtlr_groupnorm_tokens_strided": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_long, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dtlr_groupnorm_tokens": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dtlr_groupnorm_workspace_bytes": (ctypes.c_long, [c_int, c_int]),
    "dtlr_box_mlp_refine_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dtlr_box_head_refine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_long, c_int, c_int, c_void_p]),
    "dtlr_stem_pack_weights": (c_int, [c_void_p, c_void_p]),
    "dtlr_stem_conv7x7": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_stem_conv7x7_pool": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_stem_conv7x7_f32s": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dtlr_stem_conv7x7_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dtlr_maxpool3x3s2_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dtlr_preprocess_lines": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dtlr_topk_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dtlr_ctc_loss_interleaved": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "dtlr_topk_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_int, ctypes.c_long, c_int, c_int, c_void_p]),
    "dtlr_nms": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dtlr_blank_emissions": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "dtlr_blank_emissions_workspace_bytes": (ctypes.c_long, [c_int, c_int]),
    "dtlr_decode_blank": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
}


def _load(path: str) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise DTLRError(f"{path} not built: run `python -m dtlr_amd.build` (needs hipcc); "
                        "the DTLR HIP path has no CPU/PyTorch fallback")
    # torch must own the process's HIP runtime: libdtlr_hip.so NEEDs libamdhip64.so.7 and, loaded
    # first, would pull a second runtime from /opt/rocm beside torch's bundled one (kernels then
    # launch on a runtime that has no device initialised: hipErrorNoDevice).  Importing torch
    # first makes the loader resolve our dependency to the copy torch already mapped.
    import torch  # noqa: F401
    L = ctypes.CDLL(path)                  # RTLD_LOCAL: the two builds export the same symbol names and never see each other
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError if the .so is stale -> loud
        fn.restype, fn.argtypes = res, args
    return L


def lib(dtype=None) -> ctypes.CDLL:
    """libdtlr_hip.so (fp32 / fp64 / bf16 operands), or -- lib(torch.float16) -- libdtlr_hip_f16.so (fp16 operands)."""
    global _lib, _lib_f16
    if dtype is not None:
        import torch
        if dtype == torch.float16:
            if _lib_f16 is None:
                _lib_f16 = _load(LIB_PATH_F16)
            return _lib_f16
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def declared_symbols():
    return list(_SIGNATURES)


def check(code: int, what: str) -> None:
    if code != 0:
        L = lib()
        msg = L.dtlr_strerror(code).decode()
        raise DTLRError(f"{what}: {msg} (code {code}, hip error {L.dtlr_last_hip_error()})")


def ptr(t) -> int:
    return t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
