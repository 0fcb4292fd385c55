#!/usr/bin/env python3
"""Round 6 experiment: where does dtlr_gemm_k256s_multi's time go?  B = 32 encoder shapes (M = 174080 rows of 256 fp32).
Prints us per launch for: the single-slice kernel, multi with 1 / 2 / 3 / 6 symmetric plain slices, the encoder configuration
(value | offsets + bcast residual | logits 128 + bcast residual), and the same with a symmetric third slice."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import ops  # noqa: E402

B, S = 32, 5440
M = B * S
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
x = torch.randn((B, S, 256), generator=g).to(dev)
w = [(torch.randn((256, 256), generator=g) / 16).to(dev) for _ in range(6)]
img = [ops.k256s_pack(t) for t in w]
bias = torch.randn((1536,), generator=g).to(dev)
res = torch.randn((S, 384), generator=g).to(dev)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


print(f"single k256s            : {timeit(lambda: ops.gemm_k256s(x, img[0], bias[:256].contiguous())):8.1f} us")
for ns in (1, 2, 3, 4, 6):
    out = torch.empty((B, S, 256 * ns), device=dev)
    sl = [dict(wp=img[j], out=out[..., 256 * j:256 * (j + 1)], bias=bias[256 * j:256 * (j + 1)]) for j in range(ns)]
    t = timeit(lambda: ops.gemm_k256s_multi(x, sl))
    mb = (M * 1024 + M * 1024 * ns) / 1e6
    print(f"multi {ns} plain slices    : {t:8.1f} us   compulsory {mb:7.1f} MB -> {mb / t * 1e-3 * 1e3:6.2f} TB/s" .replace("TB/s", "GB/ms"))
value = torch.empty((B, S, 256), device=dev)
ow = torch.empty((B, S, 384), device=dev)
wpad = torch.cat([w[2][:128], torch.zeros((128, 256), device=dev)], 0)
enc = [dict(wp=img[0], out=value, bias=bias[:256].contiguous()),
       dict(wp=img[1], out=ow[..., :256], residual=res[:, :256]),
       dict(wp=ops.k256s_pack(wpad), out=ow[..., 256:], residual=res[:, 256:])]
print(f"encoder config (3 slices): {timeit(lambda: ops.gemm_k256s_multi(x, enc, res_rows=S)):8.1f} us")
print(f"  slices 1+2 only        : {timeit(lambda: ops.gemm_k256s_multi(x, enc[1:], res_rows=S)):8.1f} us")
print(f"  slice 0 only (posmajor): {timeit(lambda: ops.gemm_k256s_multi(x, enc[:1], res_rows=S)):8.1f} us")
print(f"  slice 1 only           : {timeit(lambda: ops.gemm_k256s_multi(x, enc[1:2], res_rows=S)):8.1f} us")
print(f"  slice 2 only           : {timeit(lambda: ops.gemm_k256s_multi(x, enc[2:], res_rows=S)):8.1f} us")
wsp = ops.split_pack(torch.cat(w[:2], 0)[:384].contiguous())
print(f"tiled resbcast (old)     : {timeit(lambda: ops.linear_resbcast(x, wsp, res)):8.1f} us")
