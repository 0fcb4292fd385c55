"""Drop-in for the reference's compiled extension module `MultiScaleDeformableAttention`
(models/dino/ops/src/vision.cpp:13-16; built by ops/setup.py), backed by libdtlr_hip.so.

    import dtlr_amd; dtlr_amd.install_dropin()      # registers sys.modules["MultiScaleDeformableAttention"]
    # ... then the reference's own ops/functions/ms_deform_attn_func.py imports and calls it unchanged

Same names, argument meaning and error behaviour as the reference wrapper
(ops/src/ms_deform_attn.h:20-61, ops/src/cuda/ms_deform_attn_cuda.cu:20-80):
  * every tensor must be contiguous and on the GPU, else RuntimeError with the reference's text;
  * CPU tensors -> "Not implemented on the CPU";
  * batch % min(batch, im2col_step) must be 0;
  * the result is a new tensor [N, Lq, M*D] enqueued on the current stream, no host sync.
fp32 and fp64 as the reference dispatches (cu:64); bf16 value (fp32 loc/attn) is an extension.
"""
from __future__ import annotations

import torch

from . import _lib

_DT = {torch.float32: _lib.DTLR_F32, torch.float64: _lib.DTLR_F64, torch.bfloat16: _lib.DTLR_BF16, torch.float16: _lib.DTLR_F16}


def _assert(cond: bool, msg: str) -> None:
    if not cond:
        raise RuntimeError(msg)


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")            # ms_deform_attn.h:38
    for t, name in ((value, "value"), (spatial_shapes, "spatial_shapes"), (level_start_index, "level_start_index"),
                    (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight")):
        _assert(t.is_contiguous(), f"{name} tensor has to be contiguous")       # cu:28-32
        _assert(t.is_cuda, f"{name} must be a CUDA tensor")                     # cu:34-38
    _assert(value.dim() == 4 and sampling_loc.dim() == 6 and attn_weight.dim() == 5, "bad tensor ranks")
    _assert(spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64,
            "spatial_shapes / level_start_index must be int64")
    N, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    _assert(tuple(sampling_loc.shape) == (N, Lq, M, L, P, 2), "sampling_loc must be [N,Lq,M,L,P,2]")
    _assert(tuple(attn_weight.shape) == (N, Lq, M, L, P), "attn_weight must be [N,Lq,M,L,P]")
    step = min(N, int(im2col_step))
    _assert(step > 0 and N % step == 0, f"batch({N}) must divide im2col_step({step})")   # cu:50-52
    if value.dtype not in _DT:
        raise RuntimeError(f'"ms_deform_attn_forward_cuda" not implemented for \'{value.dtype}\'')
    lt = torch.float32 if value.dtype in (torch.bfloat16, torch.float16) else value.dtype
    _assert(sampling_loc.dtype == lt and attn_weight.dtype == lt, "sampling_loc / attn_weight dtype mismatch")
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        code = _lib.lib(value.dtype).dtlr_msda_forward(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                            sampling_loc.data_ptr(), attn_weight.data_ptr(),
                                            N, S, M, D, L, Lq, P, _DT[value.dtype], out.data_ptr(),
                                            _lib.current_stream())
    _lib.check(code, "dtlr_msda_forward")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """The reference's backward kernels (cuh:301-921) are training-only and outside the inference
    hot path."""
    raise NotImplementedError("dtlr_amd is an inference engine: ms_deform_attn_backward is not provided")
