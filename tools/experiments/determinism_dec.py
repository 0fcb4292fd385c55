#!/usr/bin/env python3
"""Bit-reproducibility of the decoder's operators under contention (run two copies at once): each 300 times on fixed inputs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import ops
g = torch.Generator().manual_seed(0)
def R(*s, sc=1.0, dt=torch.bfloat16): return (torch.randn(s, generator=g) * sc).to(dt).cuda()
B, S, nq, M = 3, 5440, 900, 8
level_hw = [(16, 256), (8, 128), (4, 64), (2, 32)]
shapes = torch.tensor(level_hw, dtype=torch.int64).cuda()
lsi = torch.tensor([0, 4096, 5120, 5376], dtype=torch.int64).cuda()
vall = R(B, S, 1536)
value = vall[..., 256:512].unflatten(-1, (M, 32))
ow = R(B, nq, 384, sc=2.0)
ref = torch.rand((B, nq, 4, 4), generator=g).cuda() * 0.5 + 0.25
cases = {"msda_fused (decoder)": lambda: ops.msda_fused(value, shapes, lsi, ow, ref)}
qk = R(B, nq, 512); v = R(B, nq, 256)
cases["mha"] = lambda: ops.mha(qk, v, 8)
x = R(B, nq, 256); pos = R(B, nq, 256); w = R(512, 256, sc=0.05); b = torch.randn(512, generator=g).cuda()
cases["linear+a2 N512"] = lambda: ops.linear(x, w, b, a2=pos)
r4 = torch.rand((B, nq, 4), generator=g).cuda(); vr = torch.ones((B, 4, 2)).cuda()
cases["decoder_query_prep"] = lambda: ops.decoder_query_prep(r4, vr, torch.bfloat16)[1]
for name, fn in cases.items():
    ref_out = fn().clone()
    bad = 0
    for i in range(300):
        if not torch.equal(fn(), ref_out): bad += 1
    print(f"{name:26s} mismatching repeats: {bad}/300", flush=True)
