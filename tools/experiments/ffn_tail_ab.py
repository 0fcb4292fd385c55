#!/usr/bin/env python3
"""Encoder FFN call (M = 174,080 rows): whole 256-workgroup rounds on dtlr_ffn32_bf16 + the last partial round on the 16x16x32 kernel (the
engine's dispatch since round 4) against dtlr_ffn32_bf16 over all rows (2.66 rounds), interleaved in one process."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import ops
g = torch.Generator().manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 174080
x = torch.randn((M, 256), generator=g).bfloat16().cuda()
y = torch.empty_like(x)
w1 = (torch.randn((2048, 256), generator=g) / 16).bfloat16().cuda()
w2 = (torch.randn((256, 2048), generator=g) / 45).bfloat16().cuda()
b1, b2 = torch.randn(2048, generator=g).cuda() * 0.1, torch.randn(256, generator=g).cuda() * 0.1
gw, gb = torch.ones(256).cuda(), torch.zeros(256).cuda()
w2p = ops.ffn_pack_w2(w2)
w1p3, w2p3 = ops.ffn32_pack(w1, w2)
rem = M % 65536
def combo():
    ops.ffn32(x[:M - rem], w1p3, b1, w2p3, b2, gw, gb, out=y[:M - rem])
    ops.ffn_fused(x[M - rem:], w1, b1, w2p, b2, gw, gb, out=y[M - rem:])
def alone():
    ops.ffn32(x, w1p3, b1, w2p3, b2, gw, gb, out=y)
def timeit(fn, n=20):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for fn in (combo, alone): fn()
res = {"combo": [], "alone": []}
for rep in range(6):
    res["combo"].append(round(timeit(combo), 1)); res["alone"].append(round(timeit(alone), 1))
print(json.dumps({"M": M, "rem": rem, **res}))
