#!/usr/bin/env python3
"""dtlr_ffn4_bf16 (standalone build tools/experiments/ffn4v/libffn4_lagrt.so, DTLR_FFN4_LAG from the environment) and dtlr_ffn32_bf16:
time against the number of rows -- whole tile pairs per workgroup -- to separate the per-period time from the fixed part."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import ops
here = os.path.dirname(os.path.abspath(__file__))
g = torch.Generator().manual_seed(0)
d_ff = 2048
w1 = (torch.randn((d_ff, 256), generator=g) / 16).bfloat16().cuda()
w2 = (torch.randn((256, d_ff), generator=g) / 45).bfloat16().cuda()
b1, b2 = (torch.randn(d_ff, generator=g) * 0.1).cuda(), (torch.randn(256, generator=g) * 0.1).cuda()
gw, gb = torch.ones(256).cuda(), torch.zeros(256).cuda()
w1p, w2p = ops.ffn32_pack(w1, w2)
L = ctypes.CDLL(os.path.join(here, "ffn4v", os.environ.get("FFN4_LIB", "libffn4_lagrt.so")))
f = L.dtlr_ffn4_bf16
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M in [int(a) for a in os.environ.get("FFN4_M", "256,65536,131072,174080,196608,262144,393216").split(",")]:
    x = torch.randn((M, 256), generator=g).bfloat16().cuda()
    y = torch.empty_like(x)
    t4 = timeit(lambda: f(x.data_ptr(), w1p.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), gw.data_ptr(), gb.data_ptr(), 1e-5, y.data_ptr(), M, d_ff, None))
    t3 = timeit(lambda: ops.ffn32(x, w1p, b1, w2p, b2, gw, gb, out=y))
    print(f"lag {os.environ.get('DTLR_FFN4_LAG', '2'):>2s}  M={M:7d}  pairs/WG {max(1, (M + 255) // 256 / 256):5.2f}  ffn4 {t4:7.1f} us   ffn32 {t3:7.1f} us   ratio {t4 / t3:.3f}", flush=True)
