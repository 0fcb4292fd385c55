"""Synthetic text-line inputs: seeded, platform-independent (numpy PCG64).

`noise_lines`  : ImageNet-normalised noise N(0,1), the BASELINE.json bench input.
`stroke_lines` : dark pen strokes on a light background, pushed through the reference's eval
                 normalisation (x-mean)/std with mean [0.485,0.456,0.406], std [0.229,0.224,0.225]
                 (datasets/IAM.py:110-112) -- gives spatially structured activations for decode tests.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(3, 1, 1)
_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(3, 1, 1)


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def noise_lines(n: int, height: int, widths: Sequence[int] | int, seed: int = 0, start: int = 0) -> List[torch.Tensor]:
    """Lines start .. start+n-1 of the seeded stream: line i depends on (seed, i) only, so a data-parallel rank can materialise
    its shard of ONE global batch (`start` = its first global index) and any rank can recompute any other shard."""
    if isinstance(widths, int):
        widths = [widths] * n
    return [torch.from_numpy(_rng(seed * 1000003 + start + i).standard_normal((3, height, int(w))).astype(np.float32))
            for i, w in zip(range(n), widths)]


def stroke_lines(n: int, height: int, widths: Sequence[int] | int, seed: int = 0) -> List[torch.Tensor]:
    if isinstance(widths, int):
        widths = [widths] * n
    out = []
    for i, w in zip(range(n), widths):
        r = _rng(seed * 7919 + i + 17)
        w = int(w)
        img = np.full((height, w), 0.92, dtype=np.float32) + r.normal(0, 0.02, (height, w)).astype(np.float32)
        x = 6.0
        yy, xx = np.mgrid[0:height, 0:w].astype(np.float32)
        while x < w - 8:
            glyph_w = float(r.uniform(0.12, 0.32) * height)
            for _ in range(int(r.integers(2, 5))):
                x0, x1 = x + r.uniform(0, glyph_w, 2)
                y0, y1 = r.uniform(0.15 * height, 0.85 * height, 2)
                thick = float(r.uniform(0.02, 0.05) * height) + 0.7
                dx, dy = x1 - x0, y1 - y0
                den = dx * dx + dy * dy + 1e-6
                t = np.clip(((xx - x0) * dx + (yy - y0) * dy) / den, 0, 1)
                dist = np.sqrt((xx - (x0 + t * dx)) ** 2 + (yy - (y0 + t * dy)) ** 2)
                img = np.minimum(img, 0.12 + 0.8 * np.clip(dist / thick - 0.5, 0, 1))
            x += glyph_w + float(r.uniform(0.04, 0.25) * height)
        rgb = np.repeat(img[None], 3, 0)
        out.append(torch.from_numpy(((rgb - _MEAN) / _STD).astype(np.float32)))
    return out


def mixed_widths(n: int, choices: Sequence[int], seed: int = 0) -> List[int]:
    """cfg5 of BASELINE.json: widths drawn (seeded) from a set, e.g. {1536,...,2560}."""
    r = _rng(seed + 991)
    return [int(choices[int(k)]) for k in r.integers(0, len(choices), n)]


def uint8_lines(n: int, height: int, widths: Sequence[int] | int, seed: int = 0, start: int = 0) -> List[torch.Tensor]:
    """Seeded RGB uint8 line images [h, w, 3] (light paper with dark runs, like tests/util.preproc_image's odd seeds): the input of
    the eval-time preprocessing (datasets/transforms.py:78-109) in `bench.py --config latin-eval`."""
    if isinstance(widths, int):
        widths = [widths] * n
    out = []
    for i, w in zip(range(n), widths):
        g = _rng(seed * 1000003 + start + i + 555)
        base = g.integers(0, 256, (height, int(w), 3), dtype=np.uint8)
        img = np.where(g.random((height, int(w), 1)) < 0.15, base // 4, 200 + base // 5).astype(np.uint8)
        out.append(torch.from_numpy(img))
    return out
