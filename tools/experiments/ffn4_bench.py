#!/usr/bin/env python3
"""Encoder / decoder FFN calls (d_ff = 2048): the persistent two-slot kernel (dtlr_ffn4_bf16) against what the engine ran before it
(dtlr_ffn32_bf16 on whole rounds + the 16x16x32 kernel on the last partial round; the 16x16x32 kernel alone below 65536 rows)."""
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
w1 = (torch.randn((2048, 256), generator=g) / 16).bfloat16().cuda()
w2 = (torch.randn((256, 2048), generator=g) / 45).bfloat16().cuda()
b1, b2 = torch.randn(2048, generator=g).cuda() * 0.1, torch.randn(256, generator=g).cuda() * 0.1
gw, gb = torch.ones(256).cuda(), torch.zeros(256).cuda()
w2p = ops.ffn_pack_w2(w2)
w1p3, w2p3 = ops.ffn32_pack(w1, w2)
for M in [int(a) for a in sys.argv[1:]] or [174080, 131072, 43008, 28800, 5440, 900]:
    x = torch.randn((M, 256), generator=g).bfloat16().cuda()
    y = torch.empty_like(x)
    flops = 4.0 * M * 256 * 2048
    def old():
        rem = M % 65536 if M >= 65536 else M
        if M - rem:
            ops.ffn32(x[:M - rem], w1p3, b1, w2p3, b2, gw, gb, out=y[:M - rem])
        if rem:
            ops.ffn_fused(x[M - rem:], w1, b1, w2p, b2, gw, gb, out=y[M - rem:])
    t_old = timeit(old)
    t_new = timeit(lambda: ops.ffn4(x, w1p3, b1, w2p3, b2, gw, gb, out=y))
    print(json.dumps({"M": M, "before_us": round(t_old, 1), "before_tflops": round(flops / t_old / 1e6, 1), "ffn4_us": round(t_new, 1),
                      "ffn4_tflops": round(flops / t_new / 1e6, 1), "frac_of_2500": round(flops / t_new / 1e6 / 2500, 3)}), flush=True)
