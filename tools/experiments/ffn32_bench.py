#!/usr/bin/env python3
"""Encoder FFN call (M = 32 x 5440, d_ff = 2048): the 32x32x16-MFMA kernel (dtlr_ffn32_bf16) against the 16x16x32 one."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import ops

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

g = torch.Generator().manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 174080
x = torch.randn((M, 256), generator=g).bfloat16().cuda()
w1 = (torch.randn((2048, 256), generator=g) / 16).bfloat16().cuda()
w2 = (torch.randn((256, 2048), generator=g) / 45).bfloat16().cuda()
b1, b2 = torch.randn(2048, generator=g).cuda() * 0.1, torch.randn(256, generator=g).cuda() * 0.1
gw, gb = torch.ones(256).cuda(), torch.zeros(256).cuda()
w2p = ops.ffn_pack_w2(w2)
w1p3, w2p3 = ops.ffn32_pack(w1, w2)
flops = 4.0 * M * 256 * 2048
old = timeit(lambda: ops.ffn_fused(x, w1, b1, w2p, b2, gw, gb))
new = timeit(lambda: ops.ffn32(x, w1p3, b1, w2p3, b2, gw, gb))
print(json.dumps({"M": M, "ffn2_us": round(old, 1), "ffn2_tflops": round(flops / old / 1e6, 1), "ffn32_us": round(new, 1), "ffn32_tflops": round(flops / new / 1e6, 1)}))
