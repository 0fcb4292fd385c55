#!/usr/bin/env python3
"""Round 6 probe: what do plain elementwise kernels reach on this box for write-only / read-only / copy traffic (the a.R + b.W model the
weight-resident split kernels were fitted with: 0.10 us per MB read + 0.53 us per MB written)?"""
import torch

dev = "cuda:0"


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


for mb in (178, 712, 2848):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    tw = timeit(lambda: y.fill_(1.0))
    tc = timeit(lambda: y.copy_(x))
    tr = timeit(lambda: x.sum())
    ta = timeit(lambda: torch.add(x, 1.0, out=y))
    print(f"{mb:5d} MB: fill {tw:7.1f} us = {mb * 1.0486 / tw:5.2f} TB/s write | copy {tc:7.1f} us = {2 * mb * 1.0486 / tc:5.2f} TB/s r+w | "
          f"sum {tr:7.1f} us = {mb * 1.0486 / tr:5.2f} TB/s read | add {ta:7.1f} us = {2 * mb * 1.0486 / ta:5.2f} TB/s r+w")
