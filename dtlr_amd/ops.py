"""Device operators of the DTLR forward.  Every function takes/returns CUDA tensors (device
buffers) and enqueues on the current stream.

Each operator is the seam behind which a hand-written gfx950 kernel sits (libdtlr_hip.so through
dtlr_amd._lib).  Operators that do not have their HIP kernel yet are listed in `LIBRARY_BACKED`
and call the ROCm libraries through torch (rocBLAS/hipBLASLt GEMM, MIOpen conv) -- still GPU-only:
nothing here runs on the CPU and nothing imports the oracle.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

from . import MultiScaleDeformableAttention as _msda
from . import _lib

# operators still served by ROCm libraries via torch (shrinks as kernels land)
LIBRARY_BACKED = {"linear", "conv2d_nhwc", "maxpool_nhwc", "groupnorm_tokens", "layernorm", "mha", "topk", "sort"}


def require_cuda(t: torch.Tensor, what: str = "input") -> None:
    if not t.is_cuda:
        raise RuntimeError(f"dtlr_amd: {what} must live on the GPU (no CPU path; got device {t.device})")
    _lib.lib()      # raises if libdtlr_hip.so is missing


# --------------------------------------------------------------------------------------------
def linear(x, w, b=None, relu: bool = False, residual=None):
    """y = x @ w.T + b [+ReLU] [+residual]."""
    y = F.linear(x, w, b)
    if relu:
        y = F.relu(y, inplace=True)
    if residual is not None:
        y = y + residual
    return y


def layernorm(x, w, b, eps: float = 1e-5):
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps).to(x.dtype)


def conv2d_nhwc(x, w_oihw_cl, bias, stride: int, padding: int, relu: bool = False, residual=None):
    """x: [B,H,W,Cin] contiguous (NHWC); w: [Cout,Cin,kh,kw] in channels_last memory format with the
    FrozenBN scale already folded in; bias [Cout].  Returns [B,Ho,Wo,Cout] contiguous."""
    y = F.conv2d(x.permute(0, 3, 1, 2), w_oihw_cl, bias, stride=stride, padding=padding)
    y = y.permute(0, 2, 3, 1)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y, inplace=True)
    return y.contiguous()


def maxpool_nhwc(x, k: int = 3, stride: int = 2, padding: int = 1):
    y = F.max_pool2d(x.permute(0, 3, 1, 2), k, stride, padding)
    return y.permute(0, 2, 3, 1).contiguous()


def groupnorm_tokens(x, groups: int, w, b, eps: float = 1e-5):
    """GroupNorm(32, 256) over [B, T, C] tokens of one feature level: statistics per (sample,
    group) over (C/groups channels x T positions) -- models/dino/dino.py:121-134."""
    B, T, C = x.shape
    xf = x.float().reshape(B, T, groups, C // groups)
    mean = xf.mean(dim=(1, 3), keepdim=True)
    var = xf.var(dim=(1, 3), unbiased=False, keepdim=True)
    y = ((xf - mean) * torch.rsqrt(var + eps)).reshape(B, T, C) * w.float() + b.float()
    return y.to(x.dtype)


def msda(value, spatial_shapes, level_start_index, loc, attn):
    """value [N,S,M,D]; loc [N,Lq,M,L,P,2] f32; attn [N,Lq,M,L,P] f32 -> [N,Lq,M*D] (HIP kernel)."""
    return _msda.ms_deform_attn_forward(value, spatial_shapes, level_start_index, loc, attn, 64)


def mha(q, k, v, n_heads: int):
    """q,k,v [B, L, C] already projected; softmax(q k^T / sqrt(d)) v per head -> [B, L, C]."""
    B, L, C = q.shape
    hd = C // n_heads
    qh = q.view(B, L, n_heads, hd).transpose(1, 2)
    kh = k.view(B, L, n_heads, hd).transpose(1, 2)
    vh = v.view(B, L, n_heads, hd).transpose(1, 2)
    att = torch.softmax((qh.float() * (1.0 / math.sqrt(hd))) @ kh.float().transpose(-1, -2), dim=-1)
    o = (att @ vh.float()).to(q.dtype)
    return o.transpose(1, 2).reshape(B, L, C)


def topk_rows(scores, k: int):
    """Indices of the k largest per row, descending (two-stage selection,
    deformable_transformer.py:345).  Scores are always fp32."""
    return torch.topk(scores, k, dim=1)[1]
