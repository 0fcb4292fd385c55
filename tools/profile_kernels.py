#!/usr/bin/env python3
"""Standalone launches of the hand-written kernels at the bench shapes (B=32, 128x2048), for
rocprofv3 (--kernel-trace --stats, or --pmc FETCH_SIZE / WRITE_SIZE in their own passes) and for
HIP-event timing.  Prints one JSON line per kernel with the mean launch time and achieved GB/s.

    python tools/profile_kernels.py [--iters 20] [--only msda_enc_bf16,...]
A 1 GiB torch device-to-device copy is launched first as the byte-count calibration for the
FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md section HBM).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dtlr_amd import ops  # noqa: E402


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    dev = torch.device("cuda:0")
    B, M, D, L, P = args.batch, 8, 32, 4, 4
    shapes_l = [(16, 256), (8, 128), (4, 64), (2, 32)]
    S = sum(h * w for h, w in shapes_l)
    shapes = torch.as_tensor(shapes_l, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    g = torch.Generator(device="cpu").manual_seed(0)
    res = []

    def want(name):
        return not only or name in only

    if want("calib_copy"):
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev).random_()
        dst = torch.empty_like(src)
        ms = timeit(lambda: dst.copy_(src), 5)
        res.append({"kernel": "calib_copy_1GiB", "ms": ms, "bytes_read": 1 << 30, "bytes_written": 1 << 30, "GBps": 2 * (1 << 30) / ms / 1e6})
        del src, dst

    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        velem = 4 if dt == torch.float32 else 2
        value = (torch.rand((B, S, M, D), generator=g) * 2 - 1).to(dev).to(dt)
        for tag, Lq in (("enc", S), ("dec", 900)):
            # realistic encoder-like locations: reference grid + a few pixels of offset
            ow = torch.randn((B, Lq, M * L * P * 3), generator=g).to(dev)
            ow[..., : M * L * P * 2] *= 2.0
            ref2 = torch.rand((B, Lq, L, 2), generator=g).to(dev)
            ref4 = torch.cat([ref2, torch.full_like(ref2, 0.05)], -1).contiguous()
            off = ow[..., : M * L * P * 2].view(B, Lq, M, L, P, 2)
            norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
            loc = (ref2[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]).contiguous()
            aw = torch.softmax(ow[..., M * L * P * 2:].view(B, Lq, M, L * P), -1).view(B, Lq, M, L, P).contiguous()
            alg = B * (S * M * D * velem + Lq * M * L * P * 3 * 4 + Lq * M * D * velem)
            name = f"msda_{tag}_{dt_name}"
            if want(name):
                ms = timeit(lambda: ops.msda(value, shapes, lsi, loc, aw), args.iters)
                res.append({"kernel": name, "ms": ms, "algorithmic_bytes": alg, "GBps": alg / ms / 1e6})
            name = f"msda_fused_{tag}_{dt_name}"
            if want(name):
                r = ref2 if tag == "enc" else ref4
                ms = timeit(lambda: ops.msda_fused(value, shapes, lsi, ow, r), args.iters)
                res.append({"kernel": name, "ms": ms, "algorithmic_bytes": alg, "GBps": alg / ms / 1e6})
            name = f"msda_lds_{tag}_{dt_name}"
            if tag == "enc" and want(name):
                # encoder-like reference grid (pixel centres), offsets of a few pixels
                ys = [torch.linspace(0.5, h - 0.5, h) / h for h, w in shapes_l]
                xs = [torch.linspace(0.5, w - 0.5, w) / w for h, w in shapes_l]
                rp = torch.cat([torch.stack(torch.meshgrid(y, x, indexing="ij")[::-1], -1).reshape(-1, 2) for y, x in zip(ys, xs)], 0)
                refg = rp[None, :, None, :].expand(B, S, L, 2).contiguous().to(dev)
                ms = timeit(lambda: ops.msda_encoder(value, shapes_l, ow, refg), args.iters)
                res.append({"kernel": name, "ms": ms, "algorithmic_bytes": alg, "GBps": alg / ms / 1e6})
                ms = timeit(lambda: ops.msda_fused(value, shapes, lsi, ow, refg), args.iters)
                res.append({"kernel": name.replace("lds", "gather_gridref"), "ms": ms, "algorithmic_bytes": alg, "GBps": alg / ms / 1e6})
            name = f"msda_lds_{tag}_{dt_name}_engine"          # the engine's configuration: projection row in bf16 too
            if tag == "enc" and dt == torch.bfloat16 and want(name):
                ys = [torch.linspace(0.5, h - 0.5, h) / h for h, w in shapes_l]
                xs = [torch.linspace(0.5, w - 0.5, w) / w for h, w in shapes_l]
                rp = torch.cat([torch.stack(torch.meshgrid(y, x, indexing="ij")[::-1], -1).reshape(-1, 2) for y, x in zip(ys, xs)], 0)
                refg = rp[None, :, None, :].expand(B, S, L, 2).contiguous().to(dev)
                owb = ow.bfloat16()
                ms = timeit(lambda: ops.msda_encoder(value, shapes_l, owb, refg), args.iters)
                res.append({"kernel": name, "ms": ms, "algorithmic_bytes": alg, "GBps": alg / ms / 1e6})
                del owb
            del ow, loc, aw, off
        T = B * S
        x = torch.randn((T, 256), generator=g).to(dev).to(dt)
        r = torch.randn((T, 256), generator=g).to(dev).to(dt)
        w, b = torch.ones(256, device=dev), torch.zeros(256, device=dev)
        name = f"layernorm_res_{dt_name}"
        if want(name):
            ms = timeit(lambda: ops.layernorm(x, w, b, 1e-5, r), args.iters)
            alg = 3 * T * 256 * velem
            res.append({"kernel": name, "ms": ms, "algorithmic_bytes": alg, "GBps": alg / ms / 1e6})
        del x, r, value
    if want("mha_bf16"):
        qk = torch.randn((B, 900, 512), generator=g).to(dev).bfloat16()
        v = torch.randn((B, 900, 256), generator=g).to(dev).bfloat16()
        ms = timeit(lambda: ops.mha(qk, v, 8), args.iters)
        flops = B * 8 * 2 * 2 * 900 * 900 * 32
        res.append({"kernel": "mha_bf16", "ms": ms, "flops": flops, "TFLOPs": flops / ms / 1e9})
    # the MFMA GEMM at the encoder shapes vs the library (hipBLASLt through torch) on the same data
    import torch.nn.functional as F
    T = B * S
    for (N_, K_, relu) in ((256, 256, 0), (384, 256, 0), (2048, 256, 1), (256, 2048, 0)):
        for dt_name, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
            name = f"gemm_{dt_name}_M{T}_N{N_}_K{K_}"
            if only and "gemm" not in only and name not in only:
                continue
            x = torch.randn((T, K_), generator=g).to(dev).to(dt)
            w = (torch.randn((N_, K_), generator=g) / K_ ** 0.5).to(dev).to(dt)
            bb = torch.randn((N_,), generator=g).to(dev)
            flops = 2.0 * T * N_ * K_
            ms = timeit(lambda: ops.linear(x, w, bb, relu), max(5, args.iters // 2))
            res.append({"kernel": name, "ms": ms, "TFLOPs": flops / ms / 1e9})
            bl = bb.to(dt)
            ms = timeit(lambda: (F.relu(F.linear(x, w, bl), inplace=True) if relu else F.linear(x, w, bl)), max(5, args.iters // 2))
            res.append({"kernel": name + "_library", "ms": ms, "TFLOPs": flops / ms / 1e9})
            del x, w
    if "gemm_ablate" in only:
        # where does the per-slab time go?  (outputs are garbage under ablation; timing only)
        T = B * S
        for (N_, K_) in ((256, 256), (2048, 256), (256, 2048)):
            x = torch.randn((T, K_), generator=g).to(dev).bfloat16()
            w = (torch.randn((N_, K_), generator=g) / K_ ** 0.5).to(dev).bfloat16()
            flops = 2.0 * T * N_ * K_
            bb = torch.randn((N_,), generator=g).to(dev)
            for code, label in ((0, "full"), (4096, "no_stores"), (2048, "no_epilogue"), (256, "no_global_loads"), (512, "no_mfma"), (1024, "no_lds_store"),
                                (2048 + 512, "no_epi_no_mfma"), (2048 + 256 + 1024, "no_epi_no_loads_no_lds_store"), (2048 + 1792, "barriers_and_lds_reads_only")):
                os.environ["DTLR_GEMM_ABLATE"] = str(code)
                ms = timeit(lambda: ops.linear(x, w, bb, 1), 10)
                res.append({"kernel": f"gemm_ablate_bf16_N{N_}_K{K_}_{label}", "ms": ms, "TFLOPs_equiv": flops / ms / 1e9})
            os.environ["DTLR_GEMM_ABLATE"] = "0"
            del x, w
    if want("ffn_fused"):
        T = B * S
        x = torch.randn((T, 256), generator=g).to(dev).bfloat16()
        w1 = (torch.randn((2048, 256), generator=g) / 16).to(dev).bfloat16()
        w2 = (torch.randn((256, 2048), generator=g) / 45).to(dev).bfloat16()
        b1 = torch.randn((2048,), generator=g).to(dev)
        b2 = torch.randn((256,), generator=g).to(dev)
        gw, gb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
        flops = 2.0 * T * 2048 * 256 * 2
        w2p = ops.ffn_pack_w2(w2)
        ms = timeit(lambda: ops.ffn_fused(x, w1, b1, w2p, b2, gw, gb), args.iters)
        res.append({"kernel": "ffn_fused_bf16_M174080_dff2048", "ms": ms, "TFLOPs": flops / ms / 1e9})
        ms = timeit(lambda: ops.layernorm(ops.linear(ops.linear(x, w1, b1, relu=True), w2, b2), gw, gb, 1e-5, residual=x), args.iters)
        res.append({"kernel": "ffn_unfused_bf16_M174080_dff2048", "ms": ms, "TFLOPs": flops / ms / 1e9})
        del x
    if "ffn_trace" in only:
        # needs DTLR_HIP_LIB=dtlr_amd/libdtlr_hip_trace.so: per-iteration timeline of ffn2_bf16_kernel (wave 0 of the first 8 workgroups)
        import ctypes
        import numpy as np
        from dtlr_amd import _lib
        L = _lib.lib()
        if not hasattr(L, "dtlr_debug_ffn_trace"):
            raise SystemExit("ffn_trace needs the trace library (python -m dtlr_amd.build --trace; DTLR_HIP_LIB=...)")
        L.dtlr_debug_ffn_trace.restype = ctypes.c_int
        L.dtlr_debug_ffn_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
        T = B * S
        x = torch.randn((T, 256), generator=g).to(dev).bfloat16()
        w1 = (torch.randn((2048, 256), generator=g) / 16).to(dev).bfloat16()
        w2p = ops.ffn_pack_w2((torch.randn((256, 2048), generator=g) / 45).to(dev).bfloat16())
        b1, b2 = torch.randn((2048,), generator=g).to(dev), torch.randn((256,), generator=g).to(dev)
        gw, gb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
        buf = np.zeros(8 * 1024, dtype=np.uint64)
        for _ in range(3):
            ops.ffn_fused(x, w1, b1, w2p, b2, gw, gb)
        L.dtlr_debug_ffn_trace(buf.ctypes.data, 1)
        ops.ffn_fused(x, w1, b1, w2p, b2, gw, gb)
        L.dtlr_debug_ffn_trace(buf.ctypes.data, 0)
        for blk in range(8):
            ev = buf[blk * 1024:(blk + 1) * 1024]
            ev = ev[ev != 0]
            t, c = (ev >> np.uint64(8)).astype(np.int64), (ev & np.uint64(255)).astype(np.int64)
            def seg(a_, b_):                      # mean ticks from each event a_ to the next event b_
                d = [t[i + 1] - t[i] for i in range(len(c) - 1) if c[i] == a_ and c[i + 1] == b_]
                return round(float(np.mean(d)), 1) if d else None
            tops = t[c == 1]
            res.append({"kernel": f"ffn2_trace_blk{blk}", "events": int(len(ev)), "iterations": int((c == 1).sum()),
                        "ticks_per_iteration": round(float(np.mean(np.diff(tops))), 1) if len(tops) > 1 else None,
                        "wait_dma_1to2": seg(1, 2), "barrier_2to3": seg(2, 3), "body_3to5": seg(3, 5), "end_5to1": seg(5, 1),
                        "prologue_8_to_first1": int(tops[0] - t[c == 8][0]) if (c == 8).any() and len(tops) else None,
                        "tail_9to10": int(t[c == 10][0] - t[c == 9][0]) if (c == 10).any() and (c == 9).any() else None,
                        "total_8to10": int(t[c == 10][0] - t[c == 8][0]) if (c == 10).any() and (c == 8).any() else None})
        del x
    if "gemm_trace" in only:
        # needs DTLR_HIP_LIB=dtlr_amd/libdtlr_hip_trace.so (python -m dtlr_amd.build --trace): per-slab timeline of gemm_ws_kernel for
        # the first workgroups: where a slab iteration spends its cycles (loader: wait for data / LDS store / issue / barrier;
        # MFMA waves: slab / epilogue)
        import ctypes
        from dtlr_amd import _lib
        L = _lib.lib()
        if not hasattr(L, "dtlr_debug_gemm_trace"):
            raise SystemExit("gemm_trace needs the trace library (python -m dtlr_amd.build --trace; DTLR_HIP_LIB=...)")
        NB, NE = 8, 1024
        buf = (ctypes.c_ulonglong * (NB * 2 * NE))()
        T = B * S
        for (N_, K_, relu, use_res) in ((256, 256, 0, False), (256, 256, -1, False), (256, 256, 0, True)):
            x = torch.randn((T, K_), generator=g).to(dev).bfloat16()
            w = (torch.randn((N_, K_), generator=g) / K_ ** 0.5).to(dev).bfloat16()
            bb = torch.randn((N_,), generator=g).to(dev)
            rs = torch.randn((T, N_), generator=g).to(dev).bfloat16() if use_res else None
            if relu < 0:                                        # -1: no bias at all (isolates the epilogue's bias loads)
                bb, relu = None, 0
            for _ in range(3):
                ops.linear(x, w, bb, relu, rs)
            L.dtlr_debug_gemm_trace(buf, 1)
            ops.linear(x, w, bb, relu, rs)
            L.dtlr_debug_gemm_trace(buf, 0)
            import numpy as np
            arr = np.frombuffer(buf, dtype=np.uint64).reshape(NB, 2, NE).copy()
            for blk in (5,):
                for role, rname in ((0, "mfma"), (1, "loader")):
                    ev = arr[blk, role]
                    ev = ev[ev != 0]
                    t, c = (ev >> np.uint64(8)).astype(np.int64), (ev & np.uint64(0xff)).astype(np.int64)
                    if len(t) < 4:
                        continue
                    d = np.diff(t)
                    seg = {}
                    for k in range(len(d)):
                        seg.setdefault((int(c[k]), int(c[k + 1])), []).append(int(d[k]))
                    summ = {f"{a}->{b}": [len(v), int(np.median(v)), int(np.mean(v)), int(np.max(v))] for (a, b), v in sorted(seg.items())}
                    res.append({"kernel": f"gemm_trace_N{N_}_K{K_}_bias{int(bb is not None)}_res{int(use_res)}_blk{blk}_{rname}", "events": int(len(t)),
                                "span_cycles": int(t[-1] - t[0]), "segments[count,median,mean,max]": summ})
            del x, w
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
