#!/usr/bin/env python3
"""Bit-reproducibility stress: every hot operator 40 times on the same inputs (run two copies at once to add contention)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import ops
g = torch.Generator().manual_seed(0)
def R(*s, sc=1.0): return (torch.randn(s, generator=g) * sc).bfloat16().cuda()
B = 3
cases = {}
x64 = R(B, 32, 512, 64); w33 = R(64, 3, 3, 64, sc=0.05); b64 = torch.randn(64, generator=g).cuda()
os.environ.setdefault("X", "1")
cases["conv3x3 64->64 (patch or tall by env)"] = lambda: ops.conv2d_nhwc(x64, w33, b64, 1, 1, True, None)
xl = R(B * 32 * 512, 64); wl = R(64, 64, sc=0.1)
cases["linear N64 K64 (tall)"] = lambda: ops.linear(xl, wl, b64, relu=2)
x256 = R(B * 32 * 512, 256); w64 = R(64, 256, sc=0.05)
cases["linear N64 K256 (tall)"] = lambda: ops.linear(x256, w64, b64, relu=2)
wk = ops.kres_pack(w64)
cases["kres N64 K256"] = lambda: ops.gemm_kres(x256, wk, 64, b64, None, relu=True)
w256 = R(256, 64, sc=0.1); b256 = torch.randn(256, generator=g).cuda(); r256 = R(B * 32 * 512, 256)
wk2 = ops.kres_pack(w256)
cases["kres N256 K64 +res"] = lambda: ops.gemm_kres(xl, wk2, 256, b256, r256, relu=True)
cases["linear N256 K64 +res (ws)"] = lambda: ops.linear(xl, w256, b256, relu=2, residual=r256)
xt = R(B, 5440, 256); wv = R(256, 256, sc=0.05)
wvp = ops.k256_pack(wv)
cases["k256 N256"] = lambda: ops.gemm_k256(xt, wvp, 256, b256)
wo = R(384, 256, sc=0.05); res = R(5440, 384)
wob = ops.kres_pack_bcast384(wo)
cases["kres bcast384"] = lambda: ops.gemm_kres_bcast384(xt, wob, res)
w1 = R(2048, 256, sc=0.06); w2 = R(256, 2048, sc=0.02); b1 = torch.randn(2048, generator=g).cuda() * 0.1
w2p = ops.ffn_pack_w2(w2); one = torch.ones(256).cuda(); zero = torch.zeros(256).cuda()
cases["ffn_fused M16320"] = lambda: ops.ffn_fused(xt, w1, b1, w2p, zero, one, zero)
wpl = ops.proj_pack_w(wv)
cases["proj_ln M16320"] = lambda: ops.proj_ln(xt, wpl, zero, xt, one, zero)
xs = R(B, 16, 256, 128); ws2 = R(128, 3, 3, 128, sc=0.03); b128 = torch.randn(128, generator=g).cuda()
cases["conv3x3 128->128 M12288 (ws)"] = lambda: ops.conv2d_nhwc(xs, ws2, b128, 1, 1, True, None)
for name, fn in cases.items():
    ref = fn().clone()
    bad = 0
    for i in range(40):
        out = fn()
        if not torch.equal(out, ref): bad += 1
    print(f"{name:40s} mismatching repeats: {bad}/40", flush=True)
