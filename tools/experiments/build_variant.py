#!/usr/bin/env python3
"""Build libdtlr_hip_<tag>.so = the product sources with extra -D flags (experiments only; selected with DTLR_HIP_LIB).
    python tools/experiments/build_variant.py <tag> -DFOO -DBAR=2 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dtlr_amd import build as B  # noqa: E402

tag, defs = sys.argv[1], sys.argv[2:]
out = os.path.join(B.HERE, f"libdtlr_hip_{tag}.so")
B._compile_all(defs, os.path.join(B.CSRC, "var_" + tag), out, verbose=False)
print(out)
