#!/usr/bin/env python3
"""Same-process timing of the two kernels whose tile order changed in round 2 (run twice with the env knobs to A/B):
   DTLR_K256_POS_MAJOR=0|1  dtlr_gemm_k256 + broadcast residual, M = 32 x 5440, N = 384
   DTLR_TALL_XCD=0|1        3x3 conv 64 -> 64 on 32 x 32 x 512 (ResNet layer1 c2), the 256 x 64 tile kernel"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import ops


def timeit(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


g = torch.Generator().manual_seed(0)
x = torch.randn((32, 5440, 256), generator=g).bfloat16().cuda()
w = (torch.randn((384, 256), generator=g) * 0.05).bfloat16().cuda()
res = torch.randn((5440, 384), generator=g).bfloat16().cuda()
wp = ops.k256_pack(w)
out = {"env": {k: os.environ.get(k) for k in ("DTLR_K256_POS_MAJOR", "DTLR_TALL_XCD")}}
out["k256_resb_us"] = round(timeit(lambda: ops.gemm_k256(x, wp, 384, None, resid=res)), 1)
out["k256_plain_us"] = round(timeit(lambda: ops.gemm_k256(x, wp, 384, None)), 1)
xi = torch.randn((32, 32, 512, 64), generator=g).bfloat16().cuda()
wc = (torch.randn((64, 3, 3, 64), generator=g) * 0.05).bfloat16().cuda()
bc = torch.zeros(64).cuda()
out["conv3x3_l1c2_us"] = round(timeit(lambda: ops.conv2d_nhwc(xi, wc, bc, 1, 1, True, None)), 1)
print(json.dumps(out))
# where does the residual's cost come from?  same kernel with a residual that is ONE 64-row tile (always L2/L1 resident)
res64 = torch.randn((64, 384), generator=g).bfloat16().cuda()
print(json.dumps({"k256_resb_rows64_us": round(timeit(lambda: ops.gemm_k256(x, wp, 384, None, resid=res64)), 1),
                  "k256_resb_rows5440_us": round(timeit(lambda: ops.gemm_k256(x, wp, 384, None, resid=res)), 1),
                  "k256_plain_us": round(timeit(lambda: ops.gemm_k256(x, wp, 384, None)), 1)}))
wkb = ops.kres_pack_bcast384(w)
print(json.dumps({"kres_bcast384_us": round(timeit(lambda: ops.gemm_kres_bcast384(x, wkb, res)), 1)}))
