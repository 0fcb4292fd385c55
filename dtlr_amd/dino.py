"""Model-level API of the reference (seams B2/B3/builder):

  DINO.forward(samples, targets=None) -> dict     <- models/dino/dino.py:270-415
  PostProcess(num_select, nms_iou_threshold)      <- models/dino/dino.py:985-1046
  build_dino(args) -> (model, criterion, postprocessors)   <- models/dino/dino.py:1049-1194
  NestedTensor / nested_tensor_from_tensor_list   <- util/misc.py:301-397

`DINO` is an nn.Module whose parameter tree reproduces the reference state-dict keys exactly
(SURVEY.md appendix B), so `model.load_state_dict(torch.load(ckpt)["model"])` works and the
attributes evaluation.py:60-86 touches exist.  Its forward does not run those modules: the
parameters are packed once into a DTLREngine (folded BN, fused projections, NHWC) and the whole
forward is executed by the HIP/ROCm operators on the GPU.  There is no CPU path.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import nn

from . import ops
from .config import DTLRConfig
from .engine import DTLREngine
from .ms_deform_attn import MSDeformAttn


# ------------------------------------------------------------------------------- util/misc.py
class NestedTensor(object):
    def __init__(self, tensors, mask: Optional[torch.Tensor], has_padding: bool = True):
        self.tensors, self.mask = tensors, mask
        # host-side knowledge that mask is all-False lets the engine skip the value masked_fill
        # (ms_deform_attn.py:95-96) without a device sync; True = unknown / padded
        self.has_padding = has_padding

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device))

    def decompose(self):
        return self.tensors, self.mask

    @property
    def device(self):
        return self.tensors.device


def nested_tensor_from_tensor_list(tensor_list) -> NestedTensor:
    """util/misc.py:375-397: zero-pad to the batch max; mask True on padding.  One device-side
    allocation + one strided copy per image (no per-image mask writes from the host when all images
    share a shape)."""
    if isinstance(tensor_list, torch.Tensor):
        if tensor_list.ndim != 4:
            raise ValueError("not supported")
        b, c, h, w = tensor_list.shape
        return NestedTensor(tensor_list, torch.zeros((b, h, w), dtype=torch.bool, device=tensor_list.device), has_padding=False)
    if tensor_list[0].ndim != 3:
        raise ValueError("not supported")
    c = tensor_list[0].shape[0]
    h = max(int(t.shape[1]) for t in tensor_list)
    w = max(int(t.shape[2]) for t in tensor_list)
    dev, dt = tensor_list[0].device, tensor_list[0].dtype
    tensor = torch.zeros((len(tensor_list), c, h, w), dtype=dt, device=dev)
    mask = torch.ones((len(tensor_list), h, w), dtype=torch.bool, device=dev)
    for i, img in enumerate(tensor_list):
        tensor[i, :, : img.shape[1], : img.shape[2]].copy_(img)
        mask[i, : img.shape[1], : img.shape[2]] = False
    padded = any(tuple(t.shape[1:]) != (h, w) for t in tensor_list)
    return NestedTensor(tensor, mask, has_padding=padded)


# ------------------------------------------------------------------- parameter containers
class FrozenBatchNorm2d(nn.Module):
    """Buffers of models/dino/backbone.py:36-60 (folded into the convs by the engine)."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kw):
        state_dict.pop(prefix + "num_batches_tracked", None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kw)


class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            FrozenBatchNorm2d(planes * 4))


class _ResNetBody(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        inpl = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), blocks), start=1):
            layers = []
            for b in range(n):
                layers.append(_Bottleneck(inpl, planes, 2 if (b == 0 and i > 1) else 1, b == 0))
                inpl = planes * 4
            setattr(self, f"layer{i}", nn.Sequential(*layers))


class _Backbone(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.body = _ResNetBody(blocks)
        self.num_channels = [512, 1024, 2048]


class _SwinAttn(nn.Module):
    def __init__(self, dim, ws, heads):
        super().__init__()
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        self.register_buffer("relative_position_index", torch.zeros((ws * ws, ws * ws), dtype=torch.long))
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class _SwinMlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)


class _SwinBlock(nn.Module):
    def __init__(self, dim, ws, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _SwinAttn(dim, ws, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _SwinMlp(dim)


class _SwinMerge(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)


class _SwinStage(nn.Module):
    def __init__(self, dim, depth, ws, heads, downsample):
        super().__init__()
        self.blocks = nn.ModuleList(_SwinBlock(dim, ws, heads) for _ in range(depth))
        if downsample:
            self.downsample = _SwinMerge(dim)


class _SwinPatchEmbed(nn.Module):
    def __init__(self, E):
        super().__init__()
        self.proj = nn.Conv2d(3, E, kernel_size=4, stride=4)
        self.norm = nn.LayerNorm(E)


class _SwinBackbone(nn.Module):
    """Parameter container with the state-dict layout of models/dino/swin_transformer.py:435-555 (`backbone.0.*`)."""

    def __init__(self, cfg: DTLRConfig):
        super().__init__()
        sp = cfg.swin_params()
        E, ws = sp["embed_dim"], sp["window_size"]
        self.patch_embed = _SwinPatchEmbed(E)
        self.layers = nn.ModuleList(_SwinStage(E << i, sp["depths"][i], ws, sp["num_heads"][i], i < 3) for i in range(4))
        for i in cfg.return_interm_indices:
            setattr(self, f"norm{i}", nn.LayerNorm(E << i))
        self.num_channels = cfg.backbone_channels


class MLP(nn.Module):
    """models/dino/utils.py:110-122 (container)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))


class _EncoderLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg.hidden_dim
        self.self_attn = MSDeformAttn(d, cfg.num_feature_levels, cfg.nheads, cfg.enc_n_points)
        self.norm1 = nn.LayerNorm(d)
        self.linear1 = nn.Linear(d, cfg.dim_feedforward)
        self.linear2 = nn.Linear(cfg.dim_feedforward, d)
        self.norm2 = nn.LayerNorm(d)


class _DecoderLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg.hidden_dim
        self.cross_attn = MSDeformAttn(d, cfg.num_feature_levels, cfg.nheads, cfg.dec_n_points)
        self.norm1 = nn.LayerNorm(d)
        self.self_attn = nn.MultiheadAttention(d, cfg.nheads, dropout=0.0)
        self.norm2 = nn.LayerNorm(d)
        self.linear1 = nn.Linear(d, cfg.dim_feedforward)
        self.linear2 = nn.Linear(cfg.dim_feedforward, d)
        self.norm3 = nn.LayerNorm(d)


class _Stack(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)


class _Transformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg.hidden_dim
        self.d_model = d
        self.num_encoder_layers, self.num_decoder_layers = cfg.enc_layers, cfg.dec_layers
        self.num_queries = cfg.num_queries
        self.level_embed = nn.Parameter(torch.zeros(cfg.num_feature_levels, d))
        self.encoder = _Stack([_EncoderLayer(cfg) for _ in range(cfg.enc_layers)])
        self.decoder = _Stack([_DecoderLayer(cfg) for _ in range(cfg.dec_layers)])
        self.decoder.norm = nn.LayerNorm(d)
        self.decoder.ref_point_head = MLP(2 * d, d, d, 2)
        self.tgt_embed = nn.Embedding(cfg.num_queries, d)
        self.enc_output = nn.Linear(d, d)
        self.enc_output_norm = nn.LayerNorm(d)
        self.enc_out_bbox_embed = MLP(d, d, 4, 3)
        self.enc_out_class_embed = nn.Linear(d, cfg.num_classes)


# --------------------------------------------------------------------------------------- DINO
class DINO(nn.Module):
    """Cross-attention detector used as a text-line recogniser (models/dino/dino.py:49-415)."""

    def __init__(self, cfg: DTLRConfig, compute_dtype=torch.float32):
        """compute_dtype: torch.float32 (exact-fp32 MFMA: the slow parity engine), torch.bfloat16 / torch.float16 (the 16-bit
        engines) or the string "f32s" (fp32 activations, every product as three fp16 MFMAs on hi + lo halves: fp32-grade results
        at about a third of the 16-bit engines' rate -- DTLREngine(split=True))."""
        super().__init__()
        cfg.validate()
        self.cfg = cfg
        self.compute_split = compute_dtype == "f32s"
        self.compute_dtype = torch.float32 if self.compute_split else compute_dtype
        d = cfg.hidden_dim
        self.num_queries, self.num_classes, self.hidden_dim = cfg.num_queries, cfg.num_classes, d
        self.num_feature_levels, self.nheads = cfg.num_feature_levels, cfg.nheads
        self.label_enc = nn.Embedding(cfg.dn_labelbook_size + 1, d)
        proj = [nn.Sequential(nn.Conv2d(c, d, kernel_size=1), nn.GroupNorm(32, d)) for c in cfg.backbone_channels]
        proj.append(nn.Sequential(nn.Conv2d(cfg.backbone_channels[-1], d, kernel_size=3, stride=2, padding=1), nn.GroupNorm(32, d)))
        self.input_proj = nn.ModuleList(proj)
        self.backbone = nn.Sequential(_SwinBackbone(cfg) if cfg.is_swin else _Backbone(cfg.backbone_blocks))
        self.transformer = _Transformer(cfg)
        self.aux_loss = True
        self.dec_pred_class_embed_share = cfg.dec_pred_class_embed_share
        self.dec_pred_bbox_embed_share = cfg.dec_pred_bbox_embed_share
        _class_embed = nn.Linear(d, cfg.num_classes)
        _bbox_embed = MLP(d, d, 4, 3)
        self.bbox_embed = nn.ModuleList([_bbox_embed for _ in range(cfg.dec_layers)])
        self.class_embed = nn.ModuleList([_class_embed for _ in range(cfg.dec_layers)])
        self.transformer.decoder.bbox_embed = self.bbox_embed
        self.transformer.decoder.class_embed = self.class_embed
        self.two_stage_type = cfg.two_stage_type
        self._engine: Optional[DTLREngine] = None
        self.return_aux = False          # aux_outputs are skipped at inference unless asked for

    # -- engine lifecycle: any parameter change invalidates the packed weights ------------------
    def _apply(self, fn, *a, **kw):
        self._engine = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._engine = None
        sd = dict(state_dict)
        # checkpoints written through the --new_class_embedding flow carry a bare Linear under
        # transformer.decoder.class_embed.{weight,bias} (evaluation.py:81; unused by the forward).  When the caller has
        # performed that flow on THIS module (the decoder attribute is then a bare nn.Linear, evaluation.py:81) the keys
        # belong to the module and stay; otherwise they are dropped.
        if not isinstance(self.transformer.decoder.class_embed, nn.Linear):
            for k in ("transformer.decoder.class_embed.weight", "transformer.decoder.class_embed.bias"):
                sd.pop(k, None)
        return super().load_state_dict(sd, strict=strict, **kw)

    def engine(self) -> DTLREngine:
        if self._engine is None:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("dtlr_amd.DINO has no CPU path: call model.to('cuda') / model.cuda() first")
            self._engine = DTLREngine(self.cfg, self.state_dict(), dev, self.compute_dtype, split=self.compute_split)
        return self._engine

    @torch.no_grad()
    def forward(self, samples, targets: List = None, forced_topk: Optional[torch.Tensor] = None,
                return_debug: bool = False) -> Dict[str, torch.Tensor]:
        """samples: NestedTensor | Tensor[B,3,H,W] | list[Tensor[3,h,w]] (models/dino/dino.py:270-288).
        targets must be None (inference; denoising queries are training-only, dn_components.py:135-140)."""
        if targets is not None:
            raise NotImplementedError("dtlr_amd.DINO is inference-only: targets must be None")
        if self.training:
            raise RuntimeError("dtlr_amd.DINO is inference-only: call model.eval()")
        if isinstance(samples, (list, torch.Tensor)):
            samples = nested_tensor_from_tensor_list(samples)
        eng = self.engine()
        x, mask = samples.decompose()
        ops.require_cuda(x, "samples")
        out = eng.forward(x.float(), mask, forced_topk=forced_topk, want_aux=self.return_aux, return_debug=return_debug,
                          has_padding=getattr(samples, "has_padding", True))
        if not self.return_aux:
            out["aux_outputs"] = []
        return out


# -------------------------------------------------------------------------------- PostProcess
def box_cxcywh_to_xyxy(x):
    xc, yc, w, h = x.unbind(-1)
    return torch.stack([xc - 0.5 * w, yc - 0.5 * h, xc + 0.5 * w, yc + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Greedy NMS (torchvision.ops.nms semantics: suppress IoU > threshold, keep order = descending score, equal scores keep the
    lower index first) on the device: HIP kernel `dtlr_nms` (stable score sort, suppression bit-matrix and the sequential sweep
    all inside one workgroup).  No CPU path: CPU tensors raise.  The variable-length result costs one host read of the count."""
    if boxes.shape[0] == 0:
        return torch.empty(0, dtype=torch.long, device=boxes.device)
    keep, counts = ops.nms_batched(boxes[None], scores[None], iou_threshold)
    return keep[0, : int(counts[0].item())]


class PostProcess(nn.Module):
    """models/dino/dino.py:985-1046."""

    def __init__(self, num_select=100, nms_iou_threshold=-1) -> None:
        super().__init__()
        self.num_select = num_select
        self.nms_iou_threshold = nms_iou_threshold

    @torch.no_grad()
    def forward(self, outputs, target_sizes, not_to_xyxy=False, test=False):
        num_select = self.num_select
        out_logits, out_bbox = outputs["pred_logits"], outputs["pred_boxes"]
        assert len(out_logits) == len(target_sizes)
        assert target_sizes.shape[1] == 2
        ops.require_cuda(out_logits, "pred_logits")                 # device-only, like the rest of the path
        # sigmoid + flat top-k in one HIP kernel (exact radix select on the logits: sigmoid is monotone); num_select <= 8192
        topk_values, topk_indexes = ops.topk_flat(out_logits.reshape(out_logits.shape[0], -1), num_select, apply_sigmoid=True)
        scores = topk_values
        topk_boxes = topk_indexes // out_logits.shape[2]
        labels = topk_indexes % out_logits.shape[2]
        boxes = out_bbox if not_to_xyxy else box_cxcywh_to_xyxy(out_bbox)
        if test:
            assert not not_to_xyxy
            boxes[:, :, 2:] = boxes[:, :, 2:] - boxes[:, :, :2]
        boxes = torch.gather(boxes, 1, topk_boxes.unsqueeze(-1).repeat(1, 1, 4))
        img_h, img_w = target_sizes.to(boxes.device).unbind(1)
        scale_fct = torch.stack([img_w, img_h, img_w, img_h], dim=1)
        boxes = boxes * scale_fct[:, None, :]
        if self.nms_iou_threshold > 0:
            # one launch for the batch (a workgroup per image), one host read of the counts
            keep, counts = ops.nms_batched(boxes, scores, self.nms_iou_threshold)
            idx = [keep[i, :c] for i, c in enumerate(counts.tolist())]
            return [{"scores": s[i], "labels": l[i], "boxes": b[i]} for s, l, b, i in zip(scores, labels, boxes, idx)]
        return [{"scores": s, "labels": l, "boxes": b} for s, l, b in zip(scores, labels, boxes)]


# ------------------------------------------------------------------------------------ builders
def config_from_args(args) -> DTLRConfig:
    import dataclasses
    fields = {f.name for f in dataclasses.fields(DTLRConfig)}
    get = (lambda k: args[k]) if isinstance(args, dict) else (lambda k: getattr(args, k))
    has = (lambda k: k in args) if isinstance(args, dict) else (lambda k: hasattr(args, k))
    kw = {}
    for k in fields:
        if has(k):
            v = get(k)
            kw[k] = tuple(v) if isinstance(v, list) else v
    return DTLRConfig(**kw)


def build_dino(args):
    """models/dino/dino.py:1049-1194 -> (model, criterion, postprocessors).  `args` is the flat
    namespace of reference config keys (or a DTLRConfig).  criterion is None: losses/matcher are
    training-only (out of scope); the CTC blank construction lives in dtlr_amd.evaluation."""
    cfg = args if isinstance(args, DTLRConfig) else config_from_args(args)
    model = DINO(cfg)
    postprocessors = {"bbox": PostProcess(num_select=cfg.num_select, nms_iou_threshold=cfg.nms_iou_threshold)}
    return model, None, postprocessors


def build_model_main(args):
    """finetuning.py:123-131 (registry lookup of 'dino', models/registry.py:58)."""
    return build_dino(args)
