#!/usr/bin/env python3
"""Encoder MSDA (dtlr_msda_encoder_forward, bf16 engine configuration, B = 32, 128x2048 levels): time per call and fraction of
samples that take the global-memory path, as a function of the sampling-offset scale, for each query-phase form.

    python tools/msda_sweep.py [--iters 20] > profiles/r02_msda_offset_sweep.json

The LDS-window kernel is fast when a query's samples fall inside the staged column window of its tile (halo R = 8 columns per
level); the synthetic weights plant ring offsets of <= 4 px (weights._msda), a trained Latin model may attend further along the
reading direction.  Offsets here are N(0, sigma^2) PIXELS of the sampled level on both axes (sigma = 2, 8, 32, 128), the
reference grid is the encoder's (pixel centres).  Also reports max |difference| between the forms (the packed-fp16 form vs the
fp32-accumulator form) and against the gather kernel.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtlr_amd import _lib, ops  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def global_path_fraction(ow, refg, shapes_l, TW0, R):
    """Fraction of (query, head, level, point) samples that are inside the map but need a column outside the staged window of the
    query's x-tile (msda_enc.hip: cols_left_of / window bounds), computed on the host."""
    B, S, _ = ow.shape
    M, L, P = 8, 4, 4
    off = ow[..., : M * L * P * 2].float().view(B, S, M, L, P, 2)
    W0 = shapes_l[0][1]
    # query tile index: t such that the query column's centre lies in [t, t+1) * TW0 / W0
    xs = []
    for (h, w) in shapes_l:
        j = torch.arange(w, dtype=torch.float64)
        t = torch.floor(((2 * j + 1) * W0) / (2.0 * TW0 * w)).long()
        xs.append(t[None, :].expand(h, w).reshape(-1))
    tq = torch.cat(xs).to(ow.device)                                    # [S]
    glob = tot = 0
    for l, (h, w) in enumerate(shapes_l):
        lx = refg[..., l, 0][:, :, None, None] + off[:, :, :, l, :, 0] / w
        ly = refg[..., l, 1][:, :, None, None] + off[:, :, :, l, :, 1] / h
        w_im, h_im = lx * w - 0.5, ly * h - 0.5
        inside = (h_im > -1) & (w_im > -1) & (h_im < h) & (w_im < w)
        w_low = torch.floor(w_im).clamp(-1, w).long()
        w0 = w_low.clamp(0, w - 1)
        w1 = (w_low + 1).clamp(0, w - 1)
        lo = ((tq * TW0 * w) // W0 - R).clamp(min=0)[None, :, None, None]
        hi = (((tq + 1) * TW0 * w + W0 - 1) // W0 + R).clamp(max=w)[None, :, None, None]
        staged = (w0 >= lo) & (w1 < hi)
        glob += int((inside & ~staged).sum())
        tot += inside.numel()
    return glob / tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--sigmas", default="0,2,8,32,128")
    ap.add_argument("--halos", default="8", help="comma list of window halos (columns per level beyond the tile); the engine default is 8")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, M, D, L, P = args.batch, 8, 32, 4, 4
    shapes_l = [(16, 256), (8, 128), (4, 64), (2, 32)]
    S = sum(h * w for h, w in shapes_l)
    shapes = torch.as_tensor(shapes_l, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    g = torch.Generator(device="cpu").manual_seed(0)
    value = (torch.rand((B, S, M, D), generator=g) * 2 - 1).to(dev).bfloat16()
    ys = [torch.linspace(0.5, h - 0.5, h) / h for h, w in shapes_l]
    xs = [torch.linspace(0.5, w - 0.5, w) / w for h, w in shapes_l]
    rp = torch.cat([torch.stack(torch.meshgrid(y, x, indexing="ij")[::-1], -1).reshape(-1, 2) for y, x in zip(ys, xs)], 0)
    refg = rp[None, :, None, :].expand(B, S, L, 2).contiguous().to(dev)
    L_ = _lib.lib()
    alg = B * (S * M * D * 2 + S * M * L * P * 3 * 2 + S * L * 2 * 4 + S * M * D * 2)
    rows = []
    for sigma in [float(v) for v in args.sigmas.split(",")]:
        ow = torch.randn((B, S, M * L * P * 3), generator=g).to(dev)
        ow[..., : M * L * P * 2] *= sigma
        owb = ow.bfloat16()
        frac = global_path_fraction(owb, refg, shapes_l, 32, ops.MSDA_HALO)
        rec = {"offset_sigma_px": sigma, "global_path_fraction": round(frac, 5)}
        outs = {}
        # the query-phase form is fixed per process (round 4: no run-time knob in the product library): the default third form, or -- with
        # DTLR_HIP_LIB pointing at an experiment build (python -m dtlr_amd.build --instr) -- whatever DTLR_MSDA_ENC_V selects
        name = "form_" + os.environ.get("DTLR_MSDA_ENC_V", "3")
        ms = timeit(lambda: ops.msda_encoder(value, shapes_l, owb, refg), args.iters)
        outs[0] = ops.msda_encoder(value, shapes_l, owb, refg).float()
        rec[name + "_ms"] = round(ms, 4)
        rec[name + "_GBps_algorithmic"] = round(alg / ms / 1e6, 1)
        for halo in [int(v) for v in args.halos.split(",")]:
            if halo == 8:
                continue
            old = ops.MSDA_HALO
            ops.MSDA_HALO = halo
            try:
                if ops.msda_encoder_fits(shapes_l, torch.bfloat16):
                    rec[f"halo{halo}_ms"] = round(timeit(lambda: ops.msda_encoder(value, shapes_l, owb, refg), args.iters), 4)
                    rec[f"halo{halo}_global_path_fraction"] = round(global_path_fraction(owb, refg, shapes_l, 32, halo), 5)
                else:
                    rec[f"halo{halo}_ms"] = None
            finally:
                ops.MSDA_HALO = old
        gather = ops.msda_fused(value, shapes, lsi, owb, refg).float()
        rec["gather_kernel_ms"] = round(timeit(lambda: ops.msda_fused(value, shapes, lsi, owb, refg), max(3, args.iters // 4)), 4)
        rec["max_abs_diff_vs_gather"] = round((outs[0] - gather).abs().max().item(), 5)
        rows.append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
    print(json.dumps({"kernel": "msda_enc_lds_kernel<bf16,bf16>", "B": B, "S": S, "halo": ops.MSDA_HALO, "TW0": 32,
                      "algorithmic_bytes_per_launch": alg, "sweep": rows}))


if __name__ == "__main__":
    main()
