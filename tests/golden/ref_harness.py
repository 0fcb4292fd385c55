"""Import the REAL reference model (/root/reference) on CPU.  Authoring-container only.

Used by tests/golden/make_golden.py (fixture generation) and by tests marked `needs_reference`
(skipped wherever /root/reference is absent, e.g. the GPU box).  Nothing here is copied from the
reference: these are the four shims SURVEY.md section 8c lists, needed because the reference cannot
run on CPU as shipped (models/dino/dino.py:46 torch.cuda.set_device(0); the op's CPU branch throws,
ops/src/ms_deform_attn.h:38; torchvision/timm are not installed here).

The torchvision stub's ResNet-50 and NMS are THIS repo's restatement of the public definitions
(so the backbone/NMS arithmetic is "parity unpinned" against torchvision itself); everything
downstream of the backbone body -- FrozenBatchNorm2d, masks, position encoding, input_proj, the
deformable encoder/decoder, two-stage selection, heads, PostProcess, loss_CTC's blank construction
-- is the reference's own code executing.
"""
from __future__ import annotations

import os
import sys
import types
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = os.environ.get("DTLR_REFERENCE_ROOT", "/root/reference")
_BLOCKS = [3, 4, 6, 3]


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models", "dino"))


# ----------------------------------------------------------------------------- torchvision stub
class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)   # v1.5
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        o = self.relu(self.bn1(self.conv1(x)))
        o = self.relu(self.bn2(self.conv2(o)))
        o = self.bn3(self.conv3(o))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(o + idt)


class _ResNet(nn.Module):
    def __init__(self, blocks, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inpl = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), blocks), start=1):
            layers = []
            for b in range(n):
                stride = 2 if (b == 0 and i > 1) else 1
                ds = None
                if b == 0:
                    ds = nn.Sequential(nn.Conv2d(inpl, planes * 4, 1, stride=stride, bias=False), norm_layer(planes * 4))
                layers.append(_Bottleneck(inpl, planes, stride, ds, norm_layer))
                inpl = planes * 4
            setattr(self, f"layer{i}", nn.Sequential(*layers))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(2048, 1000)


def _resnet50(replace_stride_with_dilation=None, pretrained=False, norm_layer=None, **kw):
    return _ResNet(_BLOCKS, norm_layer or nn.BatchNorm2d)


class _IntermediateLayerGetter(nn.ModuleDict):
    def __init__(self, model, return_layers):
        layers = OrderedDict()
        remaining = dict(return_layers)
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = dict(return_layers)

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


def _nms(boxes, scores, iou_threshold):
    order = torch.argsort(scores, descending=True, stable=True)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    keep, dead = [], torch.zeros(len(boxes), dtype=torch.bool)
    for i in order.tolist():
        if dead[i]:
            continue
        keep.append(i)
        lt = torch.maximum(boxes[i, :2], boxes[:, :2])
        rb = torch.minimum(boxes[i, 2:], boxes[:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[:, 0] * wh[:, 1]
        dead |= inter / (area[i] + area - inter) > iou_threshold
    return torch.as_tensor(keep, dtype=torch.long)


def _install_stubs():
    if "MultiScaleDeformableAttention" in sys.modules and getattr(sys.modules["MultiScaleDeformableAttention"], "_dtlr_stub", False):
        return
    torch.cuda.set_device = lambda *a, **k: None                       # shim (1)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tv = mod("torchvision", __version__="0.16.0", _is_tracing=lambda: False)      # shim (2)
    tv.ops = mod("torchvision.ops")
    tv.ops.boxes = mod("torchvision.ops.boxes", box_area=lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]), nms=_nms)
    tv.ops.misc = mod("torchvision.ops.misc", interpolate=F.interpolate)
    tv.ops.nms = _nms
    tv.models = mod("torchvision.models", resnet50=_resnet50)
    tv.models._utils = mod("torchvision.models._utils", IntermediateLayerGetter=_IntermediateLayerGetter)
    tv.transforms = mod("torchvision.transforms")
    timm = mod("timm")                                                             # shim (3)
    timm.models = mod("timm.models")
    timm.models.layers = mod("timm.models.layers", DropPath=nn.Identity, to_2tuple=lambda x: (x, x),
                             trunc_normal_=nn.init.trunc_normal_)
    mod("MultiScaleDeformableAttention", _dtlr_stub=True)                          # shim (4)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from models.dino.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    msda = sys.modules["MultiScaleDeformableAttention"]
    msda.ms_deform_attn_forward = lambda value, shapes, lsi, loc, aw, step: ms_deform_attn_core_pytorch(value, shapes, loc, aw)


class _Args(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def load_reference_config(name: str) -> _Args:
    def load(p):
        ns = {}
        exec(compile(open(p).read(), p, "exec"), ns)
        merged = {}
        for b in ns.get("_base_", []) or []:
            merged.update(load(os.path.join(os.path.dirname(p), b)))
        merged.update({k: v for k, v in ns.items() if not k.startswith("__")})
        return merged
    cfg = _Args(load(os.path.join(REFERENCE_ROOT, "config", name)))
    cfg.device = "cpu"
    cfg.dataset_file = "IAM"
    return cfg


def build_reference_model(cfg, state_dict):
    """cfg: dtlr_amd.config.DTLRConfig -> (reference DINO module in eval mode, postprocessors,
    criterion) with `state_dict` loaded strictly."""
    global _BLOCKS
    _install_stubs()
    _BLOCKS[:] = list(cfg.backbone_blocks)
    from models.dino.dino import build_dino
    args = load_reference_config("Latin_CTC.py")
    for k in ("num_classes", "enc_layers", "dec_layers", "dim_feedforward", "num_queries", "num_select",
              "dn_labelbook_size", "hidden_dim", "nheads", "backbone"):
        args[k] = getattr(cfg, k)
    model, criterion, postprocessors = build_dino(args)
    missing, unexpected = model.load_state_dict(state_dict, strict=True)
    model.eval()
    return model, postprocessors, criterion


def reference_core_pytorch():
    _install_stubs()
    from models.dino.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    return ms_deform_attn_core_pytorch
