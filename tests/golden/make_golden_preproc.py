#!/usr/bin/env python3
"""Golden vectors for the eval-time preprocessing: tests/golden/g4_preproc.npz.

The resize arithmetic belongs to a third-party dependency of the reference (torchvision `F.resize` on a PIL image ==
Pillow `Image.resize(size, BILINEAR)`; datasets/transforms.py:108), so the vectors are produced by the installed Pillow
itself (version recorded in the file), followed by the reference's own size rule, which is imported from /root/reference
when that checkout is present (datasets/transforms.py:81-99 is a nested function: it is exercised through `resize` with a
stand-in for torchvision's F.resize that records the requested size).

Run in the authoring container:  python tests/golden/make_golden_preproc.py
Small parameter sets store the full uint8 result; the default (800, 1333) cases store SHA-256 digests.
"""
import hashlib
import os
import sys

import numpy as np
import PIL
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))

# (h, w) of the seeded source images
SMALL = [(24, 160), (31, 97), (12, 300), (40, 40), (9, 14), (64, 48)]
SMALL_PARAMS = [(20, 96), (16, 40), (48, 200), (33, None)]        # (size, max_size): down-, up-scaling, identity sizes
BIG = [(128, 2048), (64, 900), (50, 50), (200, 120), (17, 1999), (96, 1536)]


def reference_size_rule():
    """(w, h), size, max_size -> (oh, ow) through the reference's own `resize` when /root/reference is importable."""
    ref = "/root/reference"
    if not os.path.isdir(ref):
        return None
    import types
    seen = {}
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvf.resize = lambda image, size, *a, **k: seen.__setitem__("size", tuple(size)) or image
    tvt.functional = tvf
    tvt.RandomErasing = object
    tv.transforms = tvt
    for name, mod in (("torchvision", tv), ("torchvision.transforms", tvt), ("torchvision.transforms.functional", tvf)):
        sys.modules.setdefault(name, mod)
    for name in ("util.box_ops", "util.misc"):
        m = types.ModuleType(name)
        m.box_xyxy_to_cxcywh = m.interpolate = None
        sys.modules.setdefault(name, m)
    sys.modules.setdefault("util", types.ModuleType("util"))
    sys.path.insert(0, ref)
    try:
        from datasets import transforms as RT      # noqa
    except Exception as e:                         # the module drags more of torchvision than the stand-in offers
        print("reference transforms not importable here:", repr(e))
        return None

    def rule(wh, size, max_size):
        RT.resize(Image.new("RGB", wh), None, size, max_size)
        return seen["size"]
    return rule


def main():
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import dtlr_oracle as O
    from tests.util import preproc_image as image
    rule = reference_size_rule()
    out = {"pillow_version": np.array(PIL.__version__), "size_rule_from_reference": np.array(rule is not None)}
    sizes = []
    k = 0
    for i, (h, w) in enumerate(SMALL):
        img = image(h, w, 100 + i)
        for (size, max_size) in SMALL_PARAMS:
            oh, ow = O.get_size_with_aspect_ratio((w, h), size, max_size)
            if rule is not None:
                assert tuple(rule((w, h), size, max_size)) == (oh, ow), ((h, w), size, max_size)
            res = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.BILINEAR))
            out[f"s{k}_out"] = res
            sizes.append((100 + i, h, w, size, -1 if max_size is None else max_size, oh, ow))
            k += 1
    out["small_cases"] = np.array(sizes, dtype=np.int64)
    big = []
    digests = []
    for i, (h, w) in enumerate(BIG):
        img = image(h, w, 200 + i)
        oh, ow = O.get_size_with_aspect_ratio((w, h), 800, 1333)
        if rule is not None:
            assert tuple(rule((w, h), 800, 1333)) == (oh, ow)
        res = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.BILINEAR))
        digests.append(hashlib.sha256(res.tobytes()).hexdigest())
        big.append((200 + i, h, w, 800, 1333, oh, ow))
    out["big_cases"] = np.array(big, dtype=np.int64)
    out["big_sha256"] = np.array(digests)
    path = os.path.join(HERE, "g4_preproc.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; size rule from reference:", rule is not None)


if __name__ == "__main__":
    main()
