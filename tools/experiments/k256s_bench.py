#!/usr/bin/env python3
"""dtlr_gemm_k256s against the tiled split GEMM (+ dtlr_layernorm) at the encoder's token count (B = 32: 174080 rows)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from dtlr_amd import ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 174080
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((M, 256), device="cuda", generator=g)
    r = torch.randn((M, 256), device="cuda", generator=g)
    w = torch.randn((256, 256), device="cuda", generator=g) / 16
    b = torch.randn((256,), device="cuda", generator=g)
    gm, be = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
    wp, ws = ops.k256s_pack(w), ops.split_pack(w)
    t0 = timeit(lambda: ops.gemm_k256s(x, wp, b))
    t1 = timeit(lambda: ops.linear(x, ws, b))
    t2 = timeit(lambda: ops.gemm_k256s(x, wp, b, residual=r, ln_w=gm, ln_b=be))
    t3 = timeit(lambda: ops.layernorm(ops.linear(x, ws, b), gm, be, 1e-5, r))
    mk = (torch.arange(M, device="cuda") % 5 == 2)
    t4 = timeit(lambda: ops.gemm_k256s(x, wp, b, row_mask=mk, ln_w=gm, ln_b=be))
    t5 = timeit(lambda: ops.gemm_k256s(x, wp, b, row_mask=mk))
    gb = M * 256 * 4 / 1e3
    print(f"M {M}: k256s plain {t0:.1f} us ({2 * gb / t0:.0f} GB/s)  tiled {t1:.1f} us | k256s +res+LN {t2:.1f} us ({3 * gb / t2:.0f} GB/s)  tiled + LN {t3:.1f} us | masked+LN {t4:.1f} us, masked plain {t5:.1f} us")


if __name__ == "__main__":
    main()
