#!/usr/bin/env python3
"""Where does dtlr_ffn4_bf16 differ from the fp32 reference / dtlr_ffn32_bf16?  Error by tile, by 32-row wave block, by channel tile."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import ops
g = torch.Generator().manual_seed(0)
for M, d_ff in [(256, 64), (256, 2048), (1024, 2048), (1, 2048)]:
    x = torch.randn((M, 256), generator=g).bfloat16()
    w1 = (torch.randn((d_ff, 256), generator=g) / 16).bfloat16()
    w2 = (torch.randn((256, d_ff), generator=g) / 45).bfloat16()
    b1, b2 = torch.randn(d_ff, generator=g) * 0.1, torch.randn(256, generator=g) * 0.1
    gw, gb = torch.ones(256), torch.zeros(256)
    h = torch.relu(x.float() @ w1.float().t() + b1).bfloat16().float()
    pre = x.float() + h @ w2.float().t() + b2
    want = F.layer_norm(pre, (256,), gw, gb, 1e-5)
    w1p, w2p = ops.ffn32_pack(w1.cuda(), w2.cuda())
    a = [t.cuda() for t in (b1, b2, gw, gb)]
    got = ops.ffn4(x.cuda(), w1p, a[0], w2p, a[1], a[2], a[3]).float().cpu()
    old = ops.ffn32(x.cuda(), w1p, a[0], w2p, a[1], a[2], a[3]).float().cpu()
    e4, e3 = (got - want).abs(), (old - want).abs()
    print(f"M={M} d_ff={d_ff}: ffn4 max {e4.max():.4f} mean {e4.mean():.5f} | ffn32 max {e3.max():.4f} mean {e3.mean():.5f}")
    for t0 in range(0, M, 32):
        blk = e4[t0:t0 + 32]
        print(f"   rows {t0:5d}..: max {blk.max():.4f} mean {blk.mean():.5f}   by channel tile:", " ".join(f"{blk[:, c:c + 32].mean():.4f}" for c in range(0, 256, 32)))
        if t0 >= 224: break
    # is it the statistics or the values?  undo the LayerNorm scale: compare got * std + mean with pre
    mu, sd = pre.mean(-1, keepdim=True), pre.var(-1, unbiased=False, keepdim=True).add(1e-5).sqrt()
    print("   de-normalised diff (got * sd + mu - pre): max", float((got * sd + mu - pre).abs().max()), " row 0 first 8:", (got * sd + mu - pre)[0, :8].tolist())
