#!/usr/bin/env python3
"""Run standalone builds of ffn4.hip (tools/experiments/ffn4v/libffn4_<name>.so) against dtlr_ffn32_bf16: error of slot-0 / slot-1 rows, time of
the encoder call.  The libraries are built ad hoc (not tracked), e.g.
    echo 'namespace dtlr { thread_local int g_last_hip_error = 0; }' > /tmp/stub.cpp
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDTLR_EXPERIMENT [-D...] dtlr_amd/csrc/ffn4.hip /tmp/stub.cpp -o tools/experiments/ffn4v/libffn4_lagrt.so
(DTLR_EXPERIMENT makes the entry point read DTLR_FFN4_LAG: low byte = slot lag in steps, bit 8 = no LayerNorm / store slices, bit 9 =
additionally no row loads / seeding -- timing ablations whose results are garbage).  Used by tools/experiments/gpu_calls/r06_call9.sh .. call14.sh."""
import ctypes, glob, os, sys
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
xx = torch.randn((174080, 256), generator=g).bfloat16().cuda()
yy = torch.empty_like(xx)
def _old():
    ops.ffn32(xx[:131072], w1p, b1, w2p, b2, gw, gb, out=yy[:131072]); ops.ffn_fused(xx[131072:], w1, b1, ops.ffn_pack_w2(w2) if False else W2P, b2, gw, gb, out=yy[131072:])
W2P = ops.ffn_pack_w2(w2)
for _ in range(3): _old()
a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a0.record()
for _ in range(20): _old()
b0.record(); torch.cuda.synchronize()
print(f"ffn32 (131072 rows) + 16x16x32 kernel (43008 rows): {a0.elapsed_time(b0) / 20 * 1e3:.1f} us", flush=True)
for path in sorted(glob.glob(os.path.join(here, "ffn4v", "libffn4_*.so"))):
    L = ctypes.CDLL(path)
    f = L.dtlr_ffn4_bf16
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    for M in (256, 1024, 174080):
        x = torch.randn((M, 256), generator=g).bfloat16().cuda()
        want = ops.ffn32(x, w1p, b1, w2p, b2, gw, gb).float()
        worst = [0.0, 0.0]
        bad = 0
        for rep in range(5):
            y = torch.zeros_like(x)
            rc = f(x.data_ptr(), w1p.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), gw.data_ptr(), gb.data_ptr(), 1e-5, y.data_ptr(), M, d_ff, None)
            torch.cuda.synchronize()
            e = (y.float() - want).abs().nan_to_num(99.0).view(-1, 128, 256).amax((1, 2))      # per 128-row tile
            worst[0] = max(worst[0], float(e[0::2].max())); worst[1] = max(worst[1], float(e[1::2].max()))
            bad += int((e > 0.04).sum())
        t = ""
        if M >= 100000:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                f(x.data_ptr(), w1p.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), gw.data_ptr(), gb.data_ptr(), 1e-5, y.data_ptr(), M, d_ff, None)
            a.record()
            for _ in range(20):
                f(x.data_ptr(), w1p.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), gw.data_ptr(), gb.data_ptr(), 1e-5, y.data_ptr(), M, d_ff, None)
            b.record(); torch.cuda.synchronize()
            t = f"  {a.elapsed_time(b) / 20 * 1e3:.1f} us"
        print(f"{os.path.basename(path):22s} M={M:6d} rc={rc} worst |diff| slot0 {worst[0]:.4f} slot1 {worst[1]:.4f}  bad tiles (5 runs) {bad}{t}", flush=True)
