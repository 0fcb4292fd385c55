#!/usr/bin/env python3
"""Round 6 experiment (instrumented build: DTLR_HIP_LIB=dtlr_amd/libdtlr_hip_instr.so, DTLR_K256S_ORDER=0|1|2): does the tile -> workgroup
assignment of a weight-resident streaming kernel change its HBM rate?  One plain slice, one slice with a full residual, M = 174080."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import ops  # noqa: E402

B, S = 32, 5440
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
x = torch.randn((B, S, 256), generator=g).to(dev)
img = ops.k256s_pack((torch.randn((256, 256), generator=g) / 16).to(dev))
bias = torch.randn((256,), generator=g).to(dev)
r = torch.randn((B, S, 256), generator=g).to(dev)
out = torch.empty_like(x)


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


o = os.environ.get("DTLR_K256S_ORDER", "-")
t0 = timeit(lambda: ops.gemm_k256s(x, img, bias))
t1 = timeit(lambda: ops.gemm_k256s_multi(x, [dict(wp=img, out=out, bias=bias)]))
t2 = timeit(lambda: ops.gemm_k256s_multi(x, [dict(wp=img, out=out, bias=bias, residual=r.view(-1, 256))]))
print(f"order {o}: k256s {t0:7.1f} us | multi plain {t1:7.1f} us = {356.5 / t1:5.2f} TB/s | multi + full residual {t2:7.1f} us = {534.8 / t2:5.2f} TB/s")
