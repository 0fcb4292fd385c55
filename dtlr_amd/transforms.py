"""Eval-time image preprocessing -- host mirror of the reference's `datasets/transforms.py`
for the evaluation pipeline, with the pixel work done by ONE HIP launch (dtlr_preprocess_lines).

Reference pipeline (datasets/IAM.py:110-112, 225-230): `T.RandomResize([800], max_size=1333)` -> `T.ToTensor()` ->
`T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])`, then the collate `nested_tensor_from_tensor_list`
(util/misc.py:375-397).  Here: the host derives the output sizes (pure integer/float arithmetic on the image sizes,
transforms.py:81-99), uploads the uint8 pixels once, and the device resizes (Pillow's fixed-point bilinear resample, bit-exact),
scales, normalises, pads and writes the mask.

No CPU fallback: without the HIP library `preprocess_lines` raises.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .dino import NestedTensor

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
EVAL_SIZE = 800          # max(data_aug_scales), config/coco_transformer.py:1
EVAL_MAX_SIZE = 1333     # data_aug_max_size, config/coco_transformer.py:2


def get_size_with_aspect_ratio(image_size: Tuple[int, int], size: int, max_size: Optional[int] = None) -> Tuple[int, int]:
    """datasets/transforms.py:81-99.  image_size = (w, h) as PIL reports it; returns (oh, ow)."""
    w, h = image_size
    if max_size is not None:
        min_original_size = float(min((w, h)))
        max_original_size = float(max((w, h)))
        if max_original_size / min_original_size * size > max_size:
            size = int(round(max_size * min_original_size / max_original_size))
    if (w <= h and w == size) or (h <= w and h == size):
        return (h, w)
    if w < h:
        ow = size
        oh = int(size * h / w)
    else:
        oh = size
        ow = int(size * w / h)
    return (oh, ow)


def _as_hwc_u8(img) -> torch.Tensor:
    """PIL image / numpy array / torch tensor -> contiguous uint8 [h, w, 3] torch tensor (no copy when already one).
    Grey-scale input is replicated to RGB like `Image.convert("RGB")` does for mode "L" (datasets/IAM.py:86-88)."""
    if hasattr(img, "convert") and hasattr(img, "size") and not isinstance(img, (np.ndarray, torch.Tensor)):
        img = np.asarray(img.convert("RGB"))                        # a PIL image handed in by the caller
    if isinstance(img, np.ndarray):
        img = torch.from_numpy(np.ascontiguousarray(img))
    if not isinstance(img, torch.Tensor) or img.dtype != torch.uint8:
        raise TypeError("preprocess_lines: images must be uint8 [h, w, 3] (or [h, w]) arrays / tensors, or PIL images")
    if img.ndim == 2:
        img = img.unsqueeze(-1).expand(-1, -1, 3)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError(f"preprocess_lines: expected [h, w, 3], got {tuple(img.shape)}")
    return img.contiguous()


def preprocess_lines(images: Sequence, size: int = EVAL_SIZE, max_size: Optional[int] = EVAL_MAX_SIZE,
                     device="cuda", mean=IMAGENET_MEAN, std=IMAGENET_STD) -> NestedTensor:
    """List of RGB uint8 line images -> NestedTensor(tensors [B,3,Hmax,Wmax] fp32, mask [B,Hmax,Wmax] bool) on `device`,
    equal to the reference's eval transform + collate.  One host->device copy of the uint8 pixels, one kernel."""
    if len(images) == 0:
        raise ValueError("preprocess_lines: empty batch")
    device = torch.device(device)
    imgs = [_as_hwc_u8(im) for im in images]
    dims, offs, off, ratio = [], [], 0, 1.0
    for im in imgs:
        h, w = int(im.shape[0]), int(im.shape[1])
        if h == 0 or w == 0:
            raise ValueError("preprocess_lines: empty image")
        oh, ow = get_size_with_aspect_ratio((w, h), size, max_size)
        if oh <= 0 or ow <= 0:
            raise ValueError(f"preprocess_lines: image {h}x{w} resizes to an empty image ({oh}x{ow})")
        dims.append((h, w, oh, ow))
        offs.append(off)
        off += h * w * 3
        ratio = max(ratio, h / oh, w / ow)
    if all(im.is_cuda for im in imgs):
        flat = torch.cat([im.reshape(-1) for im in imgs]).to(device)
    else:
        flat = torch.cat([im.reshape(-1).cpu() for im in imgs])
        flat = flat.pin_memory().to(device, non_blocking=True) if device.type == "cuda" else flat
    dims_t = torch.tensor(dims, dtype=torch.int32).to(device)
    offs_t = torch.tensor(offs, dtype=torch.int64).to(device)
    Hc = max(d[2] for d in dims)
    Wc = max(d[3] for d in dims)
    canvas, mask = ops.preprocess_lines(flat, offs_t, dims_t, Hc, Wc, ratio, mean, std)
    return NestedTensor(canvas, mask)


class EvalTransform:
    """Callable with the reference's eval-transform semantics for a whole batch: `EvalTransform()(list_of_images)` ->
    NestedTensor.  (The reference applies its transform per item inside the Dataset and collates afterwards; the result
    is the same tensor/mask pair.)"""

    def __init__(self, size: int = EVAL_SIZE, max_size: Optional[int] = EVAL_MAX_SIZE, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.size, self.max_size, self.mean, self.std = size, max_size, mean, std

    def __call__(self, images: Sequence, device="cuda") -> NestedTensor:
        return preprocess_lines(images, self.size, self.max_size, device, self.mean, self.std)
