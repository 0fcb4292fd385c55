"""CPU ORACLE for the DTLR inference hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain single-threaded-semantics PyTorch-CPU fp32 (no custom ops, no GPU),
the algorithm of the reference's inference path: DINO.forward over text-line images -> per-query
character logits/boxes -> PostProcess / the two decoders -> CER.  Every function cites the
reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it, and only as the checker / reported baseline; nothing under
`dtlr_amd/` imports it (tests/test_host_logic.py::test_product_never_imports_oracle_and_has_no_cpu_path enforces this).

Pinning: the oracle is pinned against the REAL reference model imported in
the authoring container (`tests/golden/make_golden.py`, which loads /root/reference with the four
stubs of section 8c) on the same name-seeded weights; the resulting input/output vectors are
committed under tests/golden/ and re-checked by `tests/test_oracle_golden.py` everywhere.
Third-party arithmetic that is NOT under /root/reference -- torchvision's resnet50 topology and
`torchvision.ops.nms` -- is restated from the public definitions (ResNet-50 v1.5; greedy NMS with
IoU > threshold suppression): that part is "parity unpinned" against torchvision itself.

State-dict keys are the reference's.
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ======================================================================================
# util/misc.py
# ======================================================================================
def nested_tensor_from_tensor_list(tensor_list: Sequence[Tensor]) -> Tuple[Tensor, Tensor]:
    """util/misc.py:375-397: zero-pad to the batch max H,W; mask True on padding."""
    if isinstance(tensor_list, Tensor) and tensor_list.ndim == 4:
        tensor_list = list(tensor_list)
    assert tensor_list[0].ndim == 3
    c = tensor_list[0].shape[0]
    h = max(int(t.shape[1]) for t in tensor_list)
    w = max(int(t.shape[2]) for t in tensor_list)
    b = len(tensor_list)
    tensor = torch.zeros((b, c, h, w), dtype=tensor_list[0].dtype)
    mask = torch.ones((b, h, w), dtype=torch.bool)
    for i, img in enumerate(tensor_list):
        tensor[i, :, : img.shape[1], : img.shape[2]] = img
        mask[i, : img.shape[1], : img.shape[2]] = False
    return tensor, mask


def inverse_sigmoid(x: Tensor, eps: float = 1e-3) -> Tensor:
    """util/misc.py:575-579."""
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


def box_cxcywh_to_xyxy(x: Tensor) -> Tensor:
    """util/box_ops.py:9-13."""
    xc, yc, w, h = x.unbind(-1)
    return torch.stack([xc - 0.5 * w, yc - 0.5 * h, xc + 0.5 * w, yc + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(x: Tensor) -> Tensor:
    """util/box_ops.py:16-20."""
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


# ======================================================================================
# models/dino/backbone.py  (+ torchvision resnet50, restated from the public definition)
# ======================================================================================
def frozen_bn(x: Tensor, sd: Dict[str, Tensor], p: str) -> Tensor:
    """models/dino/backbone.py:62-72 (eps 1e-5 added before rsqrt)."""
    w = sd[p + ".weight"].reshape(1, -1, 1, 1)
    b = sd[p + ".bias"].reshape(1, -1, 1, 1)
    rv = sd[p + ".running_var"].reshape(1, -1, 1, 1)
    rm = sd[p + ".running_mean"].reshape(1, -1, 1, 1)
    scale = w * (rv + 1e-5).rsqrt()
    return x * scale + (b - rm * scale)


def resnet50_body(x: Tensor, sd: Dict[str, Tensor], blocks=(3, 4, 6, 3)) -> List[Tensor]:
    """torchvision resnet50 as called from backbone.py:118-120 (v1.5: the stride sits on the 3x3
    conv of each stage's first bottleneck; norm_layer=FrozenBatchNorm2d), returning layer2/3/4
    (IntermediateLayerGetter with return_interm_indices [1,2,3], backbone.py:84-94)."""
    b = "backbone.0.body."
    x = F.conv2d(x, sd[b + "conv1.weight"], None, stride=2, padding=3)
    x = F.relu(frozen_bn(x, sd, b + "bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = []
    for li, nblocks in enumerate(blocks, start=1):
        for bi in range(nblocks):
            p = f"{b}layer{li}.{bi}."
            stride = 2 if (bi == 0 and li > 1) else 1
            idt = x
            o = F.relu(frozen_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1"))
            o = F.relu(frozen_bn(F.conv2d(o, sd[p + "conv2.weight"], None, stride=stride, padding=1), sd, p + "bn2"))
            o = frozen_bn(F.conv2d(o, sd[p + "conv3.weight"]), sd, p + "bn3")
            if bi == 0:
                idt = frozen_bn(F.conv2d(x, sd[p + "downsample.0.weight"], None, stride=stride), sd, p + "downsample.1")
            x = F.relu(o + idt)
        if li >= 2:
            outs.append(x)
    return outs


# ======================================================================================
# models/dino/swin_transformer.py  (backbones selected by `backbone = 'swin_*'`)
# ======================================================================================
def swin_window_partition(x: Tensor, ws: int) -> Tensor:
    """swin_transformer.py:39-50: [B,H,W,C] -> [B*nW, ws, ws, C]."""
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def swin_window_reverse(windows: Tensor, ws: int, H: int, W: int) -> Tensor:
    """swin_transformer.py:53-66."""
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def swin_relative_position_index(ws: int) -> Tensor:
    """swin_transformer.py:96-106 (the registered buffer `relative_position_index`)."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij"))
    cf = torch.flatten(coords, 1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def swin_shift_mask(H: int, W: int, ws: int, shift: int) -> Tensor:
    """BasicLayer.forward, swin_transformer.py:357-376: the 0 / -100 attention mask of the shifted windows, [nW, ws*ws, ws*ws]."""
    Hp = -(-H // ws) * ws
    Wp = -(-W // ws) * ws
    img = torch.zeros((1, Hp, Wp, 1))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, h, w, :] = cnt
            cnt += 1
    mw = swin_window_partition(img, ws).view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))


def swin_window_attention(sd, p: str, x: Tensor, ws: int, nh: int, mask: Optional[Tensor]) -> Tensor:
    """WindowAttention.forward, swin_transformer.py:116-147.  x [nW*B, N, C]."""
    B_, N, C = x.shape
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(B_, N, 3, nh, C // nh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * ((C // nh) ** -0.5)
    attn = q @ k.transpose(-2, -1)
    idx = swin_relative_position_index(ws)
    bias = sd[p + "relative_position_bias_table"][idx.view(-1)].view(N, N, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, nh, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, nh, N, N)
    attn = F.softmax(attn, dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"])


def swin_block(sd, p: str, x: Tensor, H: int, W: int, ws: int, shift: int, nh: int, mask_matrix: Tensor) -> Tensor:
    """SwinTransformerBlock.forward, swin_transformer.py:191-247 (drop_path is the identity in eval)."""
    B, L, C = x.shape
    shortcut = x
    x = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5).view(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = x.shape[1], x.shape[2]
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = swin_window_partition(x, ws).view(-1, ws * ws, C)
    aw = swin_window_attention(sd, p + "attn.", xw, ws, nh, mask_matrix if shift > 0 else None)
    x = swin_window_reverse(aw.view(-1, ws, ws, C), ws, Hp, Wp)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    if pad_r > 0 or pad_b > 0:
        x = x[:, :H, :W, :].contiguous()
    x = shortcut + x.view(B, H * W, C)
    y = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    y = F.linear(F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + y


def swin_patch_merging(sd, p: str, x: Tensor, H: int, W: int) -> Tensor:
    """PatchMerging.forward, swin_transformer.py:262-288."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    if H % 2 == 1 or W % 2 == 1:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :]], -1).view(B, -1, 4 * C)
    x = F.layer_norm(x, (4 * C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
    return F.linear(x, sd[p + "reduction.weight"])


def swin_body(x: Tensor, sd: Dict[str, Tensor], sp: dict, out_indices=(1, 2, 3)) -> List[Tensor]:
    """SwinTransformer.forward (swin_transformer.py:633-673; patch_size 4, patch_norm, no ape, no dilation) -> the NCHW maps of
    `out_indices`, each after its norm{i}."""
    b = "backbone.0."
    E, depths, heads, ws = sp["embed_dim"], sp["depths"], sp["num_heads"], sp["window_size"]
    H0, W0 = x.shape[-2:]
    if W0 % 4:
        x = F.pad(x, (0, 4 - W0 % 4))
    if H0 % 4:
        x = F.pad(x, (0, 0, 0, 4 - H0 % 4))
    x = F.conv2d(x, sd[b + "patch_embed.proj.weight"], sd[b + "patch_embed.proj.bias"], stride=4)
    Wh, Ww = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (E,), sd[b + "patch_embed.norm.weight"], sd[b + "patch_embed.norm.bias"], 1e-5)
    outs = []
    for i in range(4):
        C = E << i
        mask = swin_shift_mask(Wh, Ww, ws, ws // 2)
        for j in range(depths[i]):
            x = swin_block(sd, f"{b}layers.{i}.blocks.{j}.", x, Wh, Ww, ws, 0 if j % 2 == 0 else ws // 2, heads[i], mask)
        if i in out_indices:
            o = F.layer_norm(x, (C,), sd[f"{b}norm{i}.weight"], sd[f"{b}norm{i}.bias"], 1e-5)
            outs.append(o.view(-1, Wh, Ww, C).permute(0, 3, 1, 2).contiguous())
        if i < 3:
            x = swin_patch_merging(sd, f"{b}layers.{i}.downsample.", x, Wh, Ww)
            Wh, Ww = (Wh + 1) // 2, (Ww + 1) // 2
    return outs


def interpolate_mask(mask: Tensor, size) -> Tensor:
    """backbone.py:103 / dino.py:304-307: nearest F.interpolate of the float mask, back to bool."""
    return F.interpolate(mask[None].float(), size=tuple(int(s) for s in size)).to(torch.bool)[0]


# ======================================================================================
# models/dino/position_encoding.py
# ======================================================================================
def position_embedding_sine_hw(mask: Tensor, num_pos_feats: int = 128, tH: float = 20, tW: float = 20) -> Tensor:
    """position_encoding.py:79-108 (normalize=True, scale 2*pi, eps 1e-6)."""
    not_mask = ~mask
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_tx = tW ** (2 * (dim_t // 2) / num_pos_feats)
    dim_ty = tH ** (2 * (dim_t // 2) / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_tx
    pos_y = y_embed[:, :, :, None] / dim_ty
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


# ======================================================================================
# models/dino/ops : the MSDeformAttn operator and module
# ======================================================================================
def ms_deform_attn_core(value: Tensor, spatial_shapes, sampling_locations: Tensor, attention_weights: Tensor) -> Tensor:
    """Line-by-line restatement of the CUDA forward kernel's arithmetic
    (ops/src/cuda/ms_deform_im2col_cuda.cuh:33-84 bilinear, :237-299 loop) with torch gathers:
        h_im = loc_y*H - 0.5 ; w_im = loc_x*W - 0.5 ; sampled iff -1 < h_im < H and -1 < w_im < W
        four corners, each contributing only when inside the map (zero padding)
        out[b,q,m,:] = sum_l sum_p A[b,q,m,l,p] * bilinear(...)
    which the reference states is equivalent to ms_deform_attn_core_pytorch
    (ops/functions/ms_deform_attn_func.py:41-61; ops/test.py:31-60).
    value [N,S,M,D], locations [N,Lq,M,L,P,2] (x,y), weights [N,Lq,M,L,P] -> [N,Lq,M*D]."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in (spatial_shapes.tolist() if isinstance(spatial_shapes, Tensor) else spatial_shapes)]
    out = torch.zeros((N, Lq, M, D), dtype=value.dtype)
    start = 0
    bidx = torch.arange(N).view(N, 1, 1, 1)
    midx = torch.arange(M).view(1, 1, M, 1)
    for l, (H, W) in enumerate(shapes):
        v = value[:, start:start + H * W]                       # [N, HW, M, D]
        loc = sampling_locations[:, :, :, l]                    # [N, Lq, M, P, 2]
        aw = attention_weights[:, :, :, l]                      # [N, Lq, M, P]
        w_im = loc[..., 0] * W - 0.5
        h_im = loc[..., 1] * H - 0.5
        inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        h_low = torch.floor(h_im)
        w_low = torch.floor(w_im)
        lh, lw = h_im - h_low, w_im - w_low
        hh, hw = 1 - lh, 1 - lw
        h_low, w_low = h_low.long(), w_low.long()
        h_high, w_high = h_low + 1, w_low + 1

        def corner(hi, wi, ok):
            ok = ok & inside
            idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1))           # [N,Lq,M,P]
            g = v[bidx, idx, midx]                                       # [N,Lq,M,P,D]
            return g * ok.unsqueeze(-1).to(v.dtype)

        v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
        v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
        v3 = corner(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
        v4 = corner(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))
        w1, w2, w3, w4 = hh * hw, hh * lw, lh * hw, lh * lw
        val = w1.unsqueeze(-1) * v1 + w2.unsqueeze(-1) * v2 + w3.unsqueeze(-1) * v3 + w4.unsqueeze(-1) * v4
        out += (val * aw.unsqueeze(-1)).sum(3)
        start += H * W
    return out.reshape(N, Lq, M * D)


def ms_deform_attn_core_grid_sample(value: Tensor, spatial_shapes, sampling_locations: Tensor, attention_weights: Tensor) -> Tensor:
    """The reference's OWN CPU formulation, ms_deform_attn_core_pytorch (ops/functions/ms_deform_attn_func.py:41-61): per level,
    F.grid_sample(bilinear, zeros, align_corners=False) on 2 * loc - 1, then the attention-weighted sum.  Numerically equal to
    `ms_deform_attn_core` above (tests/test_oracle_golden.py); this is the form whose SPEED is the reference's CPU speed, so
    bench.py's cpu_baseline leg selects it (MSDA_CORE = "grid_sample")."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in (spatial_shapes.tolist() if isinstance(spatial_shapes, Tensor) else spatial_shapes)]
    value_list = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for l, (H, W) in enumerate(shapes):
        v = value_list[l].flatten(2).transpose(1, 2).reshape(N * M, D, H, W)
        g = grids[:, :, :, l].transpose(1, 2).flatten(0, 1)                       # [N*M, Lq, P, 2]
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))   # [N*M, D, Lq, P]
    aw = attention_weights.transpose(1, 2).reshape(N * M, 1, Lq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()


# which core ms_deform_attn_module uses: "gather" (the kernel-arithmetic restatement, default: bit-level checks) or "grid_sample"
MSDA_CORE = "gather"


def msda_sampling_locations(reference_points: Tensor, sampling_offsets: Tensor, spatial_shapes: Tensor, n_points: int) -> Tensor:
    """ops/modules/ms_deform_attn.py:102-111."""
    if reference_points.shape[-1] == 2:
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1).to(sampling_offsets.dtype)
        return reference_points[:, :, None, :, None, :] + sampling_offsets / normalizer[None, None, None, :, None, :]
    if reference_points.shape[-1] == 4:
        return reference_points[:, :, None, :, None, :2] + sampling_offsets / n_points * reference_points[:, :, None, :, None, 2:] * 0.5
    raise ValueError("Last dim of reference_points must be 2 or 4")


def ms_deform_attn_module(sd, p, query, reference_points, input_flatten, spatial_shapes, padding_mask,
                          n_heads=8, n_levels=4, n_points=4) -> Tensor:
    """ops/modules/ms_deform_attn.py:78-126 (MSDeformAttn.forward)."""
    N, Lq, C = query.shape
    _, S, _ = input_flatten.shape
    value = F.linear(input_flatten, sd[p + ".value_proj.weight"], sd[p + ".value_proj.bias"])
    if padding_mask is not None:
        value = value.masked_fill(padding_mask[..., None], 0.0)
    value = value.view(N, S, n_heads, C // n_heads)
    off = F.linear(query, sd[p + ".sampling_offsets.weight"], sd[p + ".sampling_offsets.bias"]).view(N, Lq, n_heads, n_levels, n_points, 2)
    aw = F.linear(query, sd[p + ".attention_weights.weight"], sd[p + ".attention_weights.bias"]).view(N, Lq, n_heads, n_levels * n_points)
    aw = F.softmax(aw, -1).view(N, Lq, n_heads, n_levels, n_points)
    loc = msda_sampling_locations(reference_points, off, spatial_shapes, n_points)
    core = ms_deform_attn_core_grid_sample if MSDA_CORE == "grid_sample" else ms_deform_attn_core
    out = core(value, spatial_shapes, loc, aw)
    return F.linear(out, sd[p + ".output_proj.weight"], sd[p + ".output_proj.bias"])


# ======================================================================================
# models/dino/utils.py
# ======================================================================================
def gen_encoder_output_proposals(memory: Tensor, padding_mask: Tensor, spatial_shapes) -> Tuple[Tensor, Tensor]:
    """models/dino/utils.py:15-64 (learnedwh=None)."""
    N, S, C = memory.shape
    proposals = []
    cur = 0
    for lvl, (H, W) in enumerate(spatial_shapes.tolist()):
        m = padding_mask[:, cur:cur + H * W].view(N, H, W, 1)
        valid_H = torch.sum(~m[:, :, 0, 0], 1)
        valid_W = torch.sum(~m[:, 0, :, 0], 1)
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H, dtype=torch.float32),
                                torch.linspace(0, W - 1, W, dtype=torch.float32), indexing="ij")
        grid = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1)
        scale = torch.cat([valid_W.unsqueeze(-1), valid_H.unsqueeze(-1)], 1).view(N, 1, 1, 2)
        grid = (grid.unsqueeze(0).expand(N, -1, -1, -1) + 0.5) / scale
        wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
        proposals.append(torch.cat((grid, wh), -1).view(N, -1, 4))
        cur += H * W
    prop = torch.cat(proposals, 1)
    valid = ((prop > 0.01) & (prop < 0.99)).all(-1, keepdim=True)
    prop = torch.log(prop / (1 - prop))
    prop = prop.masked_fill(padding_mask.unsqueeze(-1), float("inf"))
    prop = prop.masked_fill(~valid, float("inf"))
    mem = memory.masked_fill(padding_mask.unsqueeze(-1), 0.0)
    mem = mem.masked_fill(~valid, 0.0)
    return mem, prop


def mlp(sd, p, x: Tensor, num_layers: int) -> Tensor:
    """models/dino/utils.py:110-122."""
    for i in range(num_layers):
        x = F.linear(x, sd[f"{p}.layers.{i}.weight"], sd[f"{p}.layers.{i}.bias"])
        if i < num_layers - 1:
            x = F.relu(x)
    return x


def gen_sineembed_for_position(pos_tensor: Tensor) -> Tensor:
    """models/dino/utils.py:141-167 (4-d boxes -> [y|x|w|h] x 128)."""
    scale = 2 * math.pi
    dim_t = torch.arange(128, dtype=torch.float32)
    dim_t = 10000 ** (2 * (dim_t // 2) / 128)

    def emb(c):
        e = (c * scale)[:, :, None] / dim_t
        return torch.stack((e[:, :, 0::2].sin(), e[:, :, 1::2].cos()), dim=3).flatten(2)

    px, py = emb(pos_tensor[:, :, 0]), emb(pos_tensor[:, :, 1])
    if pos_tensor.size(-1) == 2:
        return torch.cat((py, px), dim=2)
    pw, ph = emb(pos_tensor[:, :, 2]), emb(pos_tensor[:, :, 3])
    return torch.cat((py, px, pw, ph), dim=2)


# ======================================================================================
# models/dino/deformable_transformer.py
# ======================================================================================
def get_valid_ratio(mask: Tensor) -> Tensor:
    """deformable_transformer.py:239-246 -> (w, h)."""
    _, H, W = mask.shape
    valid_H = torch.sum(~mask[:, :, 0], 1)
    valid_W = torch.sum(~mask[:, 0, :], 1)
    return torch.stack([valid_W.float() / W, valid_H.float() / H], -1)


def encoder_reference_points(spatial_shapes, valid_ratios: Tensor) -> Tensor:
    """deformable_transformer.py:479-492."""
    lst = []
    for lvl, (H, W) in enumerate(spatial_shapes.tolist()):
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=torch.float32),
                                torch.linspace(0.5, W - 0.5, W, dtype=torch.float32), indexing="ij")
        ry = ry.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
        rx = rx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
        lst.append(torch.stack((rx, ry), -1))
    ref = torch.cat(lst, 1)
    return ref[:, :, None] * valid_ratios[:, None]


def layer_norm(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def encoder_layer(sd, p, src, pos, ref, spatial_shapes, padding_mask, cfg) -> Tensor:
    """deformable_transformer.py:804-823 (post-norm, dropout 0)."""
    src2 = ms_deform_attn_module(sd, p + ".self_attn", src + pos, ref, src, spatial_shapes, padding_mask,
                                 cfg.nheads, cfg.num_feature_levels, cfg.enc_n_points)
    src = layer_norm(sd, p + ".norm1", src + src2)
    ff = F.linear(F.relu(F.linear(src, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                  sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return layer_norm(sd, p + ".norm2", src + ff)


def multihead_self_attention(sd, p, q_in: Tensor, v_in: Tensor, n_heads: int) -> Tensor:
    """nn.MultiheadAttention(256, 8) as used at deformable_transformer.py:847,904-907:
    q = k = tgt + query_pos, v = tgt; batch-first restatement ([B, nq, C]); scale 1/sqrt(head_dim);
    no masks in eval."""
    B, Lq, C = q_in.shape
    W, bias = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(q_in, W[:C], bias[:C])
    k = F.linear(q_in, W[C:2 * C], bias[C:2 * C])
    v = F.linear(v_in, W[2 * C:], bias[2 * C:])
    hd = C // n_heads
    q = q.view(B, Lq, n_heads, hd).transpose(1, 2)
    k = k.view(B, Lq, n_heads, hd).transpose(1, 2)
    v = v.view(B, Lq, n_heads, hd).transpose(1, 2)
    att = torch.softmax((q * (1.0 / math.sqrt(hd))) @ k.transpose(-1, -2), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, Lq, C)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def decoder_layer(sd, p, tgt, query_pos, ref_input, memory, spatial_shapes, padding_mask, cfg) -> Tensor:
    """deformable_transformer.py:981-997 with module_seq ['sa','ca','ffn'] (sa 882-923, ca 925-959,
    ffn 876-880); batch-first restatement of the seq-first reference."""
    t2 = multihead_self_attention(sd, p + ".self_attn", tgt + query_pos, tgt, cfg.nheads)
    tgt = layer_norm(sd, p + ".norm2", tgt + t2)
    t2 = ms_deform_attn_module(sd, p + ".cross_attn", tgt + query_pos, ref_input, memory, spatial_shapes, padding_mask,
                               cfg.nheads, cfg.num_feature_levels, cfg.dec_n_points)
    tgt = layer_norm(sd, p + ".norm1", tgt + t2)
    ff = F.linear(F.relu(F.linear(tgt, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                  sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return layer_norm(sd, p + ".norm3", tgt + ff)


def transformer_forward(sd, cfg, srcs, masks, poss, forced_topk: Optional[Tensor] = None,
                        resume: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
    """DeformableTransformer.forward (deformable_transformer.py:257-429), inference branch
    (refpoint_embed=None, tgt=None, attn_mask=None).  `forced_topk` [B,nq] overrides the two-stage
    selection indices: used by parity tests to separate rounding-induced rank swaps of near-tied
    scores (a discrete effect present between ANY two fp32 implementations) from arithmetic error.
    `resume` = the dict an earlier call on the SAME inputs returned: the encoder output is taken from it
    (memory, masks, shapes are functions of the inputs only) and only the selection + decoder are
    recomputed -- what makes a teacher-forced re-run of many lines affordable for the checkers."""
    t = "transformer."
    if resume is not None:
        memory, mask_flatten = resume["memory"], resume["mask_flatten"]
        spatial_shapes, valid_ratios = resume["spatial_shapes"], resume["valid_ratios"]
    else:
        src_f, mask_f, pos_f, shapes = [], [], [], []
        for lvl, (src, mask, pos) in enumerate(zip(srcs, masks, poss)):
            bs, c, h, w = src.shape
            shapes.append((h, w))
            src_f.append(src.flatten(2).transpose(1, 2))
            mask_f.append(mask.flatten(1))
            pos_f.append(pos.flatten(2).transpose(1, 2) + sd[t + "level_embed"][lvl].view(1, 1, -1))
        src_flatten = torch.cat(src_f, 1)
        mask_flatten = torch.cat(mask_f, 1)
        lvl_pos = torch.cat(pos_f, 1)
        spatial_shapes = torch.as_tensor(shapes, dtype=torch.long)
        valid_ratios = torch.stack([get_valid_ratio(m) for m in masks], 1)

        # ---- encoder (494-580) ----
        ref = encoder_reference_points(spatial_shapes, valid_ratios)
        memory = src_flatten
        for n in range(cfg.enc_layers):
            memory = encoder_layer(sd, f"{t}encoder.layers.{n}", memory, lvl_pos, ref, spatial_shapes, mask_flatten, cfg)

    # ---- two-stage query selection (320-363) ----
    output_memory, output_proposals = gen_encoder_output_proposals(memory, mask_flatten, spatial_shapes)
    output_memory = layer_norm(sd, t + "enc_output_norm",
                               F.linear(output_memory, sd[t + "enc_output.weight"], sd[t + "enc_output.bias"]))
    enc_class = F.linear(output_memory, sd[t + "enc_out_class_embed.weight"], sd[t + "enc_out_class_embed.bias"])
    enc_coord = mlp(sd, t + "enc_out_bbox_embed", output_memory, 3) + output_proposals
    topk_scores = enc_class.max(-1)[0]
    topk_idx = torch.topk(topk_scores, cfg.num_queries, dim=1)[1] if forced_topk is None else forced_topk
    refpoint_undetach = torch.gather(enc_coord, 1, topk_idx.unsqueeze(-1).repeat(1, 1, 4))
    init_box_proposal = torch.gather(output_proposals, 1, topk_idx.unsqueeze(-1).repeat(1, 1, 4)).sigmoid()
    tgt_undetach = torch.gather(output_memory, 1, topk_idx.unsqueeze(-1).repeat(1, 1, cfg.hidden_dim))
    bs = memory.shape[0]
    tgt = sd[t + "tgt_embed.weight"][None].repeat(bs, 1, 1)          # embed_init_tgt=True (354-355)

    # ---- decoder (652-766), batch-first ----
    reference_points = refpoint_undetach.sigmoid()
    ref_points = [reference_points]
    intermediate = []
    vr4 = torch.cat([valid_ratios, valid_ratios], -1)                 # [B, L, 4]
    output = tgt
    for n in range(cfg.dec_layers):
        ref_input = reference_points[:, :, None] * vr4[:, None]       # [B, nq, L, 4]
        query_sine = gen_sineembed_for_position(ref_input[:, :, 0, :])
        query_pos = mlp(sd, t + "decoder.ref_point_head", query_sine, 2)
        output = decoder_layer(sd, f"{t}decoder.layers.{n}", output, query_pos, ref_input, memory,
                               spatial_shapes, mask_flatten, cfg)
        delta = mlp(sd, f"bbox_embed.{n}", output, 3)
        new_ref = (delta + inverse_sigmoid(reference_points)).sigmoid()
        reference_points = new_ref
        ref_points.append(new_ref)
        intermediate.append(layer_norm(sd, t + "decoder.norm", output))
    return dict(hs=intermediate, references=ref_points, hs_enc=tgt_undetach, ref_enc=refpoint_undetach.sigmoid(),
                init_box_proposal=init_box_proposal, topk_idx=topk_idx, topk_scores=topk_scores,
                memory=memory, spatial_shapes=spatial_shapes, valid_ratios=valid_ratios, mask_flatten=mask_flatten)


# ======================================================================================
# models/dino/dino.py
# ======================================================================================
def input_proj_level(sd, l: int, x: Tensor, stride: int = 1, padding: int = 0) -> Tensor:
    """dino.py:115-136: Conv + GroupNorm(32, 256)."""
    x = F.conv2d(x, sd[f"input_proj.{l}.0.weight"], sd[f"input_proj.{l}.0.bias"], stride=stride, padding=padding)
    return F.group_norm(x, 32, sd[f"input_proj.{l}.1.weight"], sd[f"input_proj.{l}.1.bias"], 1e-5)


@torch.no_grad()
def dino_forward(sd: Dict[str, Tensor], cfg, samples, mask: Optional[Tensor] = None,
                 forced_topk: Optional[Tensor] = None, return_debug: bool = False,
                 resume: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
    """DINO.forward (models/dino/dino.py:270-415), eval, targets=None.
    `samples`: [B,3,H,W] tensor (+ optional explicit mask) or list of [3,h,w] tensors.
    `resume` = the `_debug` dict of an earlier call on the same samples (optionally row-sliced with
    `resume_rows`): backbone and encoder are not recomputed, only selection + decoder + heads."""
    if resume is not None:
        return _dino_heads(sd, cfg, transformer_forward(sd, cfg, None, None, None, forced_topk, resume=resume), return_debug, {})
    if mask is None:
        x, mask = nested_tensor_from_tensor_list(samples)
    else:
        x = samples
    if getattr(cfg, "is_swin", False):
        feats = swin_body(x, sd, cfg.swin_params(), tuple(cfg.return_interm_indices))     # backbone.py:172-205
    else:
        feats = resnet50_body(x, sd, cfg.backbone_blocks)
    srcs, masks, poss = [], [], []
    for l, f in enumerate(feats):
        m = interpolate_mask(mask, f.shape[-2:])
        srcs.append(input_proj_level(sd, l, f))
        masks.append(m)
        poss.append(position_embedding_sine_hw(m, cfg.hidden_dim // 2, cfg.pe_temperatureH, cfg.pe_temperatureW))
    l = len(feats)
    src = input_proj_level(sd, l, feats[-1], stride=2, padding=1)      # dino.py:299-301
    m = interpolate_mask(mask, src.shape[-2:])                        # from the FULL-RES mask (304-307)
    srcs.append(src)
    masks.append(m)
    poss.append(position_embedding_sine_hw(m, cfg.hidden_dim // 2, cfg.pe_temperatureH, cfg.pe_temperatureW))

    tr = transformer_forward(sd, cfg, srcs, masks, poss, forced_topk)
    return _dino_heads(sd, cfg, tr, return_debug, dict(srcs=srcs, masks=masks, poss=poss, feats=feats))


def resume_rows(debug: Dict[str, Tensor], rows) -> Dict[str, Tensor]:
    """the `resume` argument of dino_forward for a subset of the lines of an earlier call"""
    rows = torch.as_tensor(list(rows), dtype=torch.long)
    return dict(memory=debug["memory"][rows], mask_flatten=debug["mask_flatten"][rows],
                spatial_shapes=debug["spatial_shapes"], valid_ratios=debug["valid_ratios"][rows])


def _dino_heads(sd, cfg, tr, return_debug, extra) -> Dict[str, Tensor]:
    """dino.py:339-415: class / box heads of every decoder layer, the interm outputs."""
    hs, reference = tr["hs"], tr["references"]
    coords, classes = [], []
    for n in range(cfg.dec_layers):                                    # dino.py:339-354
        coords.append((mlp(sd, f"bbox_embed.{n}", hs[n], 3) + inverse_sigmoid(reference[n])).sigmoid())
        classes.append(F.linear(hs[n], sd[f"class_embed.{n}.weight"], sd[f"class_embed.{n}.bias"]))
    out = {"pred_logits": classes[-1], "pred_boxes": coords[-1],
           "aux_outputs": [{"pred_logits": a, "pred_boxes": b} for a, b in zip(classes[:-1], coords[:-1])]}
    t = "transformer."
    interm_class = F.linear(tr["hs_enc"], sd[t + "enc_out_class_embed.weight"], sd[t + "enc_out_class_embed.bias"])
    out["interm_outputs"] = {"pred_logits": interm_class, "pred_boxes": tr["ref_enc"]}
    out["interm_outputs_for_matching_pre"] = {"pred_logits": interm_class, "pred_boxes": tr["init_box_proposal"]}
    out["dn_meta"] = None
    if return_debug:
        out["_debug"] = dict(tr, **extra)
    return out


# ======================================================================================
# PostProcess + decoders + metrics
# ======================================================================================
def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """torchvision.ops.nms (public definition; not under /root/reference -> parity unpinned):
    greedy, descending score, suppress boxes with IoU > threshold; returns kept indices sorted by
    descending score.  Called from models/dino/dino.py:1029-1033."""
    order = torch.argsort(scores, descending=True, stable=True)
    x1, y1, x2, y2 = boxes.unbind(-1)
    area = (x2 - x1) * (y2 - y1)
    keep = []
    suppressed = torch.zeros(len(boxes), dtype=torch.bool)
    for i in order.tolist():
        if suppressed[i]:
            continue
        keep.append(i)
        xx1 = torch.maximum(x1[i], x1)
        yy1 = torch.maximum(y1[i], y1)
        xx2 = torch.minimum(x2[i], x2)
        yy2 = torch.minimum(y2[i], y2)
        inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
        iou = inter / (area[i] + area - inter)
        suppressed |= iou > iou_threshold
    return torch.as_tensor(keep, dtype=torch.long)


@torch.no_grad()
def post_process(outputs, target_sizes: Tensor, num_select: int, nms_iou_threshold: float = -1,
                 not_to_xyxy: bool = False, test: bool = False) -> List[Dict[str, Tensor]]:
    """PostProcess.forward (models/dino/dino.py:985-1046)."""
    out_logits, out_bbox = outputs["pred_logits"], outputs["pred_boxes"]
    assert len(out_logits) == len(target_sizes)
    assert target_sizes.shape[1] == 2
    prob = out_logits.sigmoid()
    topk_values, topk_indexes = torch.topk(prob.view(out_logits.shape[0], -1), num_select, dim=1)
    scores = topk_values
    topk_boxes = topk_indexes // out_logits.shape[2]
    labels = topk_indexes % out_logits.shape[2]
    boxes = out_bbox if not_to_xyxy else box_cxcywh_to_xyxy(out_bbox)
    if test:
        assert not not_to_xyxy
        boxes[:, :, 2:] = boxes[:, :, 2:] - boxes[:, :, :2]
    boxes = torch.gather(boxes, 1, topk_boxes.unsqueeze(-1).repeat(1, 1, 4))
    img_h, img_w = target_sizes.unbind(1)
    scale_fct = torch.stack([img_w, img_h, img_w, img_h], dim=1)
    boxes = boxes * scale_fct[:, None, :]
    if nms_iou_threshold > 0:
        idx = [nms(b, s, nms_iou_threshold) for b, s in zip(boxes, scores)]
        return [{"scores": s[i], "labels": l[i], "boxes": b[i]} for s, l, b, i in zip(scores, labels, boxes, idx)]
    return [{"scores": s, "labels": l, "boxes": b} for s, l, b in zip(scores, labels, boxes)]


@torch.no_grad()
def decode_nms(outputs, TH: float, NM: float) -> List[List[int]]:
    """evaluation.convert_output_to_pred, NMS branch (evaluation.py:94-115), applied per sample
    (the reference runs batch 1): num_select=900 (all queries' top entries), size (1,1),
    keep score > TH, order by box cx."""
    res = []
    B, nq, _ = outputs["pred_logits"].shape
    for b in range(B):
        one = {"pred_logits": outputs["pred_logits"][b:b + 1], "pred_boxes": outputs["pred_boxes"][b:b + 1]}
        o = post_process(one, torch.tensor([[1.0, 1.0]]), num_select=900 if nq >= 900 else nq, nms_iou_threshold=NM)[0]
        boxes = box_xyxy_to_cxcywh(o["boxes"])
        sel = o["scores"] > TH
        order = torch.sort(boxes[sel][:, 0], descending=False)[1]
        res.append([int(i) for i in o["labels"].long()[sel][order]])
    return res


@torch.no_grad()
def blank_probabilities(outputs, eps: float) -> Tensor:
    """Shared front of evaluation.py:116-151 and SetCriterion.loss_CTC (dino.py:466-502):
    sort queries by cx, sigmoid, build the blank channel.  eps = 0.03/C (evaluation.py:141) or
    0.003 (dino.py:491)."""
    logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
    _, idx = torch.sort(boxes[:, :, 0])
    p = torch.gather(logits, 1, idx.unsqueeze(-1).expand(-1, -1, logits.shape[-1])).sigmoid()
    new = torch.zeros((p.shape[0], p.shape[1], p.shape[2] + 1))
    new[:, :, 1:] = p
    mask = p.sum(-1) < 1 - eps
    new[:, :, 0][mask] = 1 - p[mask].sum(-1)
    mask = ~mask
    new[:, :, 0][mask] = eps
    new[:, :, 1:][mask] = (1 - eps) * p[mask] / p[mask].sum(-1).unsqueeze(-1)
    return new


@torch.no_grad()
def decode_blank(outputs, eps: Optional[float] = None) -> List[List[int]]:
    """Blank/argmax decoder: evaluation.py:116-158 (which reads batch index 0 only; applied here to
    every sample independently) == engine.convert_output_to_pred (engine.py:511-530): argmax over
    [blank | classes], drop blanks, NO repeat collapse."""
    C = outputs["pred_logits"].shape[-1]
    new = blank_probabilities(outputs, 0.03 / C if eps is None else eps)
    pred = new.max(-1)[1]
    return [[int(i) - 1 for i in row[row != 0]] for row in pred]


def levenshtein(s1, s2) -> int:
    """evaluation.py:309-326 / engine.py:607-624 (== editdistance.eval, the un-vendored C++ package
    used at evaluation.py:519,524)."""
    if len(s1) < len(s2):
        return levenshtein(s2, s1)
    if len(s2) == 0:
        return len(s1)
    prev = list(range(len(s2) + 1))
    for i, c1 in enumerate(s1):
        cur = [i + 1]
        for j, c2 in enumerate(s2):
            cur.append(min(prev[j + 1] + 1, cur[j] + 1, prev[j] + (c1 != c2)))
        prev = cur
    return prev[-1]


def character_error_rate_engine(pred, gt) -> float:
    """engine.py:594-633 (returns 1 when either side is empty)."""
    cer = levenshtein(pred, gt) / max(len(gt), 1)
    if len(gt) == 0 or len(pred) == 0:
        cer = 1
    return cer


def process_pred_string(s: str) -> str:
    """evaluation.py:430-450."""
    s = s.replace("B B C", "BBC")
    s = s.replace("I T V", "ITV")
    s = s.replace("  ", " ")
    s = s.replace(" -", "-")
    s = s.replace("- ", "-")
    s = s.replace(" .", ".")
    s = s.replace(" ,", ",")
    s = re.sub(r"(\d), (\d)", r"\1,\2", s)
    s = s.replace(""" '""", "'")
    s = s.replace("""' """, "'")
    s = re.sub(r"(?<=\S)€(?=\S)", " € ", s)
    s = re.sub(r"(?<!\.)\.\.(?!\.)", ".", s)
    s = s.replace(",,", ",")
    return s


def cumulative_cer(gt_strings: Sequence[str], pred_strings: Sequence[str], normalise: bool = True) -> Tuple[float, List[float]]:
    """evaluation.py:517-529,547,653-656: running sum(dist)/sum(len) appended per sample; the
    reported number is the MEAN of that running series (a quirk of the reference)."""
    dists, lens, series = [], [], []
    for g, p in zip(gt_strings, pred_strings):
        if normalise:
            g, p = process_pred_string(g), process_pred_string(p)
        dists.append(levenshtein(g, p))
        lens.append(len(g))
        series.append(sum(dists) / sum(lens))
    return (sum(series) / len(series) if series else 0.0), series


def word_error_rate(predicted_words, gt_words) -> float:
    """evaluation.py:358-396: Levenshtein over word lists / max(len(gt_words), 1).  NOTE the harness calls it as
    word_error_rate(gt_split, pred_split) (evaluation.py:533-535, 546-549), i.e. with the roles swapped: the reported WER is
    normalised by the number of PREDICTED words.  Callers here keep that argument order."""
    return levenshtein(predicted_words, gt_words) / max(len(gt_words), 1)


def split_labels_into_words(labels, charset) -> List[List[int]]:
    """evaluation.py:400-412: split a label sequence at the charset's space character; empty words are dropped."""
    space = charset.index(" ")
    words, word = [], []
    for label in labels:
        if label == space:
            if word:
                words.append(word)
                word = []
        else:
            word.append(label)
    if word:
        words.append(word)
    return words


def process_gt_string(s: str) -> str:
    """evaluation.py:414-428."""
    s = s.replace("B B C", "BBC")
    s = s.replace("I T V", "ITV")
    s = s.replace(" -", "-")
    s = s.replace("- ", "-")
    s = s.replace(" -", "-")
    s = s.replace("- ", "-")
    s = s.replace(" .", ".")
    s = s.replace(" ,", ",")
    s = s.replace(""" '""", "'")
    s = s.replace("""' """, "'")
    s = re.sub(r"(\d), (\d)", r"\1,\2", s)
    s = re.sub(r"(?<=\S)€(?=\S)", " € ", s)
    return s


def character_error_rate_with_impact(pred, gt, impact: Dict[int, int]):
    """evaluation.py:162-210: (cer, impact, div).  The "impact" bookkeeping counts, for EVERY cell of the DP table whose two
    characters differ, the predicted character (the dict is not an error attribution, just that count); kept as is because the
    harness writes it to dict_char.json.  The reference raises on an empty gt (its helper returns a bare int there)."""
    if len(gt) == 0:
        raise TypeError("character_error_rate_with_impact: empty ground truth (the reference fails to unpack here)")
    for p_ in pred:
        n = sum(1 for g_ in gt if g_ != p_)
        if n:
            impact[int(p_)] = impact.get(int(p_), 0) + n
    dist = levenshtein(pred, gt)
    return dist / max(len(gt), 1), impact, max(len(gt), 1)


def compute_wa(gt, pred) -> float:
    """evaluation.py:212-238 compute_WA: positions i < min(len) with pred[i] == gt[i], over max(len(gt), 1)."""
    if len(pred) == 0:
        return 0 / max(len(gt), 1)
    return sum(1 for a, b in zip(gt, pred) if a == b) / max(len(gt), 1)


def compute_edit_operations(s1, s2) -> Tuple[int, int, int]:
    """evaluation.py:239-281: (insertions, deletions, substitutions) of one optimal alignment; the backtrace prefers a
    substitution, then a deletion, then an insertion."""
    m, n = len(s1), len(s2)
    dp = [[0] * (n + 1) for _ in range(m + 1)]
    for i in range(m + 1):
        for j in range(n + 1):
            if i == 0:
                dp[i][j] = j
            elif j == 0:
                dp[i][j] = i
            elif s1[i - 1] == s2[j - 1]:
                dp[i][j] = dp[i - 1][j - 1]
            else:
                dp[i][j] = 1 + min(dp[i - 1][j], dp[i][j - 1], dp[i - 1][j - 1])
    i, j = m, n
    ins = dele = sub = 0
    while i > 0 and j > 0:
        if s1[i - 1] == s2[j - 1]:
            i, j = i - 1, j - 1
        elif dp[i][j] == dp[i - 1][j - 1] + 1:
            sub += 1
            i, j = i - 1, j - 1
        elif dp[i][j] == dp[i - 1][j] + 1:
            dele += 1
            i -= 1
        elif dp[i][j] == dp[i][j - 1] + 1:
            ins += 1
            j -= 1
    return ins + j, dele + i, sub


def compute_cr(gt, pred) -> float:
    """evaluation.py:283-290 compute_CR (Chinese "correct rate"): (len(gt) - deletions - substitutions) / len(gt)."""
    _, dele, sub = compute_edit_operations(gt, pred)
    return (len(gt) - (dele + sub)) / len(gt)


def format_string_for_wer(s: str) -> List[str]:
    """engine.py:487-494: punctuation becomes its own word, runs of blanks/newlines collapse, split on spaces."""
    s = re.sub(r'''([\[\]{}/\()"'&+*=<>?.;:,!\-—_€#%°])''', r' \1 ', s)     # NB: the backslash is NOT in the class (it escapes '(')
    s = re.sub('([ \n])+', " ", s).strip()
    return s.split(" ")


def compute_wer_engine(pred_labels: Sequence[Sequence[int]], target_labels: Sequence[Sequence[int]], charset, mode_chr: bool = True):
    """engine.py:543-593 compute_wer for already decoded label sequences (duplicate=False): returns (sum of per-sample WER,
    sum of per-sample engine CER).  WER = word-level edit distance of the format_string_for_wer lists / number of gt words;
    '¬' is dropped from both strings.  mode_chr: charset entries are characters (True) or code points (False: chr(int(x)))."""
    wer = cer = 0.0
    for pred, tgt in zip(pred_labels, target_labels):
        cer += character_error_rate_engine(list(pred), [int(t) for t in tgt])
        ts = [charset[int(t)] for t in tgt]
        ps = [charset[int(p_)] for p_ in pred]
        if not mode_chr:
            ts, ps = [chr(int(t)) for t in ts], [chr(int(p_)) for p_ in ps]
        gt_words = format_string_for_wer("".join(ts).replace("¬", ""))
        pr_words = format_string_for_wer("".join(ps).replace("¬", ""))
        wer += levenshtein(gt_words, pr_words) / len(gt_words)
    return wer, cer


# ======================================================================================
# ngram/prediction_helpers.py -- n-gram re-scoring around a CTC beam decoder
# ======================================================================================
def ngram_new_pred_logits(output, multiply_pred_logits_by: float = 1.0) -> Tensor:
    """prediction_helpers.py:5-46 (get_new_pred_logits)."""
    logits, boxes = output["pred_logits"], output["pred_boxes"]
    _, idx = torch.sort(boxes[:, :, 0])
    p = torch.gather(logits, 1, idx.unsqueeze(-1).expand(-1, -1, logits.shape[-1])).sigmoid() * multiply_pred_logits_by
    new = torch.zeros((p.shape[0], p.shape[1], p.shape[2] + 1))
    new[:, :, 1:] = p
    eps = 0.003
    mask = p.sum(-1) < 1 - eps
    new[:, :, 0][mask] = 1 - p[mask].sum(-1)
    mask = ~mask
    new[:, :, 0][mask] = eps
    new[:, :, 1:][mask] = (1 - eps) * p[mask] / p[mask].sum(-1).unsqueeze(-1)
    return new


def ngram_first_non_0(label_list):
    """prediction_helpers.py:117-121 (UnboundLocalError on an empty list, like the reference)."""
    for e in label_list:
        if e > 0:
            break
    return e                                          # noqa: F821 - deliberately unbound for an empty list


def ngram_input_split_indices(new_pred_logits, ngram_charset, indices_to_ignore, no_uppercase_words=True, no_digits=False, no_dash=True):
    """prediction_helpers.py:124-173."""
    model_labels = new_pred_logits[0].argmax(-1)
    mask = (model_labels[:, None] == torch.tensor(indices_to_ignore)[None, :]).any(-1)
    split_indices = [-1] + torch.where(mask)[0].tolist() + [len(new_pred_logits[0])]
    if not (no_uppercase_words or no_digits):
        return split_indices, split_indices
    clean = []
    for i in range(len(split_indices) - 1):
        try:
            first = ngram_first_non_0(model_labels[split_indices[i] + 1: split_indices[i + 1] - 1].tolist())
        except UnboundLocalError:
            continue
        if first == 0:
            continue
        if no_uppercase_words and ngram_charset[first].isupper():
            continue
        if no_digits and ngram_charset[first].isdigit():
            continue
        elif no_dash and (ngram_charset.index("-") in model_labels[split_indices[i] + 1: split_indices[i + 1]].tolist()):
            continue
        else:
            clean.append(split_indices[i])
    clean.append(len(new_pred_logits[0]))
    return split_indices, clean


def ngram_word_per_word_pred(new_pred_logits, ctc_decoder, indices_to_ignore, charset) -> str:
    """prediction_helpers.py:49-74."""
    mask = (new_pred_logits[0].argmax(-1)[:, None] == torch.tensor(indices_to_ignore)[None, :]).any(-1)
    split_indices = [-1] + torch.where(mask)[0].tolist() + [len(new_pred_logits[0])]
    characs = []
    model_labels = new_pred_logits[0].argmax(-1)
    for i in range(len(split_indices) - 1):
        if split_indices[i] < split_indices[i + 1] - 1:
            word = new_pred_logits[0][split_indices[i] + 1: split_indices[i + 1]][None, :, :]
            characs += ctc_decoder(word)[0][0].words
        if split_indices[i + 1] < len(new_pred_logits[0]):
            characs += charset[model_labels[split_indices[i + 1]] - 1]
    return "".join(characs)


def ngram_word_per_word_pred_2(new_pred_logits, ctc_decoder, indices_to_ignore, ngram_charset, no_uppercase_words, no_digits, no_dash) -> str:
    """prediction_helpers.py:176-224."""
    model_labels = new_pred_logits[0].argmax(-1)
    split_indices, clean = ngram_input_split_indices(new_pred_logits, ngram_charset, indices_to_ignore, no_uppercase_words, no_digits, no_dash)
    characs = []
    max_added = -1
    for i in range(len(split_indices) - 1):
        a, b = split_indices[i], split_indices[i + 1]
        if (a in split_indices[1:]) and (a > max_added):
            characs += ngram_charset[model_labels[a]]
            max_added = a
        if (a < b) and (a in clean):
            characs += ctc_decoder(new_pred_logits[0][a + 1: b][None, :, :])[0][0].words
            max_added = max(b - 1, max_added)
        else:
            word = model_labels[a + 1: b]
            characs += [ngram_charset[c] for c in word[word > 0]]
            max_added = max(b - 1, max_added)
        if (b in split_indices[:-1]) and (b > max_added):
            characs += ngram_charset[model_labels[b]]
            max_added = b
    return "".join(characs)


# ======================================================================================
# datasets/transforms.py -- eval-time preprocessing
# ======================================================================================
def get_size_with_aspect_ratio(image_size: Tuple[int, int], size: int, max_size: Optional[int] = None) -> Tuple[int, int]:
    """datasets/transforms.py:81-99: image_size = (w, h) -> (oh, ow); short side -> `size`, capped so that the long side
    stays <= max_size."""
    w, h = image_size
    if max_size is not None:
        mn, mx = float(min(w, h)), float(max(w, h))
        if mx / mn * size > max_size:
            size = int(round(max_size * mn / mx))
    if (w <= h and w == size) or (h <= w and h == size):
        return (h, w)
    if w < h:
        return (int(size * h / w), size)
    return (size, int(size * w / h))


_PIL_PRECISION_BITS = 32 - 8 - 2


def _pil_bilinear_coeffs(in_size: int, out_size: int):
    """Pillow `precompute_coeffs` + `normalize_coeffs_8bpc` (src/libImaging/Resample.c) for the BILINEAR (triangle) filter over
    the full box [0, in_size): per output index (first tap, list of fixed-point weights).  Third-party arithmetic (torchvision
    `F.resize` on a PIL image == `Image.resize(size[::-1], BILINEAR)`, datasets/transforms.py:108): pinned against the installed
    Pillow in tests/test_oracle_golden.py and through the fixtures made by tests/golden/make_golden_preproc.py."""
    scale = in_size / out_size                     # doubles, exactly as the C code
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ss = 1.0 / filterscale
    out = []
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ws = []
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w = 1.0 - a if a < 1.0 else 0.0
            ws.append(w)
            ww += w
        ks = []
        for w in ws:
            if ww != 0.0:
                w = w / ww
            ks.append(int(-0.5 + w * (1 << _PIL_PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << _PIL_PRECISION_BITS)))
        out.append((xmin, ks))
    return out


def pil_resize_bilinear_u8(img, oh: int, ow: int):
    """uint8 [h, w, C] numpy -> uint8 [oh, ow, C]: Pillow's two-pass resample (horizontal into a uint8 image, then vertical),
    accumulators seeded with half an LSB, `>> 22`, clamp to [0, 255].  A pass whose size does not change is skipped
    (Resample.c `need_horizontal` / `need_vertical`); equal sizes return a copy (Image.resize)."""
    import numpy as np
    h, w, _ = img.shape
    cur = img
    if ow != w:
        co = _pil_bilinear_coeffs(w, ow)
        tmp = np.empty((h, ow, img.shape[2]), dtype=np.uint8)
        for xx, (x0, ks) in enumerate(co):
            acc = np.full((h, img.shape[2]), 1 << (_PIL_PRECISION_BITS - 1), dtype=np.int64)
            for i, k in enumerate(ks):
                acc += cur[:, x0 + i, :].astype(np.int64) * k
            tmp[:, xx, :] = np.clip(acc >> _PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
        cur = tmp
    if oh != h:
        co = _pil_bilinear_coeffs(h, oh)
        tmp = np.empty((oh, cur.shape[1], img.shape[2]), dtype=np.uint8)
        for yy, (y0, ks) in enumerate(co):
            acc = np.full((cur.shape[1], img.shape[2]), 1 << (_PIL_PRECISION_BITS - 1), dtype=np.int64)
            for i, k in enumerate(ks):
                acc += cur[y0 + i, :, :].astype(np.int64) * k
            tmp[yy, :, :] = np.clip(acc >> _PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
        cur = tmp
    return cur.copy() if cur is img else cur


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def preprocess_line(img, size: int = 800, max_size: int = 1333) -> Tensor:
    """One RGB uint8 [h, w, 3] line image -> normalised fp32 [3, oh, ow]: RandomResize([800], max_size=1333) -> ToTensor ->
    Normalize (datasets/IAM.py:110-112, 225-230; datasets/transforms.py:78-109, 247-249, 552-559)."""
    h, w, _ = img.shape
    oh, ow = get_size_with_aspect_ratio((w, h), size, max_size)
    r = pil_resize_bilinear_u8(img, oh, ow)
    t = torch.from_numpy(r).permute(2, 0, 1).to(torch.float32).div(255)                 # F.to_tensor
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(3, 1, 1)
    return t.sub(mean).div(std)                                                         # F.normalize


def preprocess_lines(images, size: int = 800, max_size: int = 1333) -> Tuple[Tensor, Tensor]:
    """List of RGB uint8 images -> (padded batch [B, 3, Hmax, Wmax] fp32, mask [B, Hmax, Wmax] bool): the per-item transform
    above followed by the collate of util/misc.py:375-397."""
    return nested_tensor_from_tensor_list([preprocess_line(im, size, max_size) for im in images])


# ======================================================================================
# models/dino/dino.py:457-551 -- SetCriterion.loss_CTC forward value
# ======================================================================================
def loss_ctc(outputs, target_labels: Sequence[Sequence[int]], eps: float = 0.003, filler: float = 1e-5) -> Tensor:
    """The evaluation-time CTC loss value of engine.evaluate_CTC (engine.py:381): queries in reading order, sigmoid, blank
    channel (dino.py:474-502 == blank_probabilities with eps 0.003), a filler step [1, 1e-5, ...] after every query
    (:505-519, sequence length 2 nq), `nn.CTCLoss(blank=0, zero_infinity=True, reduction="mean")` on the log of that
    against labels + 1 (:520-544).  Returns the scalar loss (fp32, CPU)."""
    probs = blank_probabilities(outputs, eps)                                   # [B, nq, C + 1]
    B, nq, C1 = probs.shape
    blank_rows = torch.zeros_like(probs) + filler
    blank_rows[:, :, 0] = 1
    padded = torch.zeros((B, 2 * nq, C1), dtype=probs.dtype)
    padded[:, ::2, :] = probs
    padded[:, 1::2, :] = blank_rows
    lengths = torch.tensor([len(t) for t in target_labels], dtype=torch.int64)
    tt = torch.zeros((B, int(lengths.max().item()) if B else 0))
    for i, t in enumerate(target_labels):
        tt[i, : len(t)] = torch.as_tensor(list(t), dtype=tt.dtype) + 1
    return F.ctc_loss(torch.log(padded.permute(1, 0, 2)), tt, torch.full((B,), 2 * nq, dtype=torch.int64), lengths,
                      blank=0, reduction="mean", zero_infinity=True)
