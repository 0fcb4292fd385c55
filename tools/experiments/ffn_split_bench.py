#!/usr/bin/env python3
"""Time the standalone builds of dtlr_amd/csrc/ffn_split.hip (tools/experiments/libffn_split_<tag>.so: -DFS_STANDALONE plus a look-ahead /
ablation macro) at the encoder and decoder shapes of the bench.  DBG builds compute garbage: timing only."""
import ctypes
import glob
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dtlr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
d_ff = 2048
w1 = (torch.randn(d_ff, 256, generator=g) / 16).to(dev)
w2 = (torch.randn(256, d_ff, generator=g) / 45).to(dev)
b1, b2 = torch.randn(d_ff, generator=g).to(dev) * 0.5, torch.randn(256, generator=g).to(dev) * 0.3
lw, lb = (torch.randn(256, generator=g) * 0.2 + 1).to(dev), torch.randn(256, generator=g).to(dev) * 0.1
wp = ops.ffn_split_pack(w1, w2)
ref = {}
for M in (174080, 28800):
    x = torch.randn(M, 256, generator=g).to(dev)
    ref[M] = (x, ops.ffn_split(x, wp, b1, b2, lw, lb))
st = torch.cuda.current_stream().cuda_stream
for path in sorted(glob.glob(os.path.join(ROOT, "tools", "experiments", "libffn_split_*.so"))):
    tag = os.path.basename(path)[len("libffn_split_"):-3]
    L = ctypes.CDLL(path)
    fn = L.dtlr_ffn_split
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    row = {"variant": tag}
    for M, (x, want) in ref.items():
        y = torch.empty_like(x)
        call = lambda: fn(x.data_ptr(), wp.data_ptr(), b1.data_ptr(), b2.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-5, y.data_ptr(), M, d_ff, st)
        for _ in range(3):
            assert call() == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            call()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        row[f"M{M}_us"] = round(us, 1)
        row[f"M{M}_tflops_x3"] = round(3 * 4.0 * M * 256 * d_ff / us / 1e6, 0)
        if not tag.startswith("D"):
            row[f"M{M}_equal_product_kernel"] = bool(torch.equal(y, want))
    print(json.dumps(row), flush=True)
