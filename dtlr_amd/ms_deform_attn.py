"""Operator-level API of the reference, backed by the HIP kernel:
  MSDeformAttnFunction  <- models/dino/ops/functions/ms_deform_attn_func.py:21-38
  MSDeformAttn          <- models/dino/ops/modules/ms_deform_attn.py:30-126
Same constructor/forward signatures and state-dict keys; inference only (backward raises)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function

from . import MultiScaleDeformableAttention as MSDA
from . import ops


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        return MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                           sampling_locations, attention_weights, ctx.im2col_step)

    @staticmethod
    def backward(ctx, grad_output):
        return MSDA.ms_deform_attn_backward(None, None, None, None, None, grad_output, ctx.im2col_step)


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)

    @torch.no_grad()
    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        """Same contract as ops/modules/ms_deform_attn.py:78-126 (shapes in its docstring)."""
        ops.require_cuda(query, "query")
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        assert (input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum() == Len_in
        value = ops.linear(input_flatten, self.value_proj.weight, self.value_proj.bias)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)
        off = ops.linear(query, self.sampling_offsets.weight, self.sampling_offsets.bias).view(
            N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
        aw = ops.linear(query, self.attention_weights.weight, self.attention_weights.bias).view(
            N, Len_q, self.n_heads, self.n_levels * self.n_points)
        aw = F.softmax(aw, -1).view(N, Len_q, self.n_heads, self.n_levels, self.n_points)
        if reference_points.shape[-1] == 2:
            normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = reference_points[:, :, None, :, None, :2] + off / self.n_points * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(reference_points.shape[-1]))
        out = MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                         loc.contiguous(), aw.contiguous(), self.im2col_step)
        return ops.linear(out, self.output_proj.weight, self.output_proj.bias)
