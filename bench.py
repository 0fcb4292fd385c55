#!/usr/bin/env python3
"""bench.py -- DTLR inference hot path on MI355X: text-lines/sec on synthetic 128x2048 crops.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of 32 synthetic lines per GPU, inputs already
resident in HBM: DINO forward (backbone -> encoder -> two-stage -> decoder -> heads) + the blank
decoder producing per-line label records (+ ONE RCCL all-gather of those records when N > 1).
Workload = BASELINE.json configs[1]: Latin model (C=166), bf16, bs=32, 128x2048, random-init
name-seeded weights (no checkpoint ships with the reference).  Rank 0 prints ONE JSON line.

roofline     : the kernel class with the largest share of the timed region, measured live with HIP
               events recorded on the launch stream around every launch of the timed steps:
               MFMA-bound classes (GEMM / implicit-GEMM conv, fused FFN): achieved = algorithmic flops per
               launch (2 M N K; 4 M d d_ff) / mean launch duration vs the dense bf16 MFMA peak;
               HBM-bound (deformable sampling, encoder call Lq = S = 5440 per line): algorithmic bytes
               per launch (SURVEY.md 8d / DESIGN.md) / mean launch duration vs the HBM peak, with the
               PMC-measured traffic.  `roofline_by_kernel` lists every class.
cpu_baseline : the CPU oracle (oracle/dtlr_oracle.py, a port pinned to the reference through
               tests/golden) timed on the host cores of this box on a bounded sample (rank 0, N=1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

T_START = time.perf_counter()


def log(msg: str) -> None:
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak (no sparsity); fp32 MFMA (16x16x4_f32) class: 157
MFMA_PEAK_F32_TFLOPS = 157.0


def msda_algorithmic_bytes_per_line(S: int, Lq: int, value_elem: int, M=8, D=32, L=4, P=4, ref_dim=2) -> int:
    """Compulsory bytes of one MSDA call for one line ("all operands in the element size e"): read
    value once, read the offsets + attention logits of every (head, level, point) once (3 numbers each: the fused kernel
    takes them as the raw projection row, in the engine dtype), read the fp32 reference points, write out."""
    return S * M * D * value_elem + Lq * M * L * P * 3 * value_elem + Lq * L * ref_dim * 4 + Lq * M * D * value_elem


def cpu_baseline(n_lines: int, height: int, width: int, repeats: int, threads: int = 0):
    """Bounded sample (target 10-30 s of CPU work): the oracle forward + blank decode on n_lines
    synthetic lines, fp32.  threads = 0: the oracle's torch-CPU ops stop scaling (and regress) far below
    the box's core count, so a few thread counts are tried and the FASTEST is reported, with the
    thread count actually used in `cores`."""
    from dtlr_amd import synth, weights
    from dtlr_amd.config import DTLRConfig
    from oracle import dtlr_oracle as O      # the reported CPU baseline (a port); never the product path
    cfg = DTLRConfig.latin()
    sd = weights.synthetic_state_dict(cfg, seed=0)
    imgs = synth.noise_lines(n_lines, height, width, seed=123)
    ncpu = os.cpu_count() or 8

    def run():
        t0 = time.perf_counter()
        out = O.dino_forward(sd, cfg, imgs)
        O.decode_blank(out)
        return time.perf_counter() - t0

    cands = [threads] if threads > 0 else sorted({min(16, ncpu), min(32, ncpu), min(64, ncpu)})
    best = None
    spent = 0.0
    for th in cands:
        if spent > 25.0 and best is not None:
            break
        torch.set_num_threads(th)
        warm = run()
        ts = []
        for _ in range(repeats):
            ts.append(run())
            if sum(ts) + warm > 12.0:
                break
        spent += warm + sum(ts)
        med = sorted(ts)[len(ts) // 2]
        log(f"cpu_baseline: {th} threads: warm-up {warm:.2f}s, timed {['%.2f' % t for t in ts]}")
        if best is None or med < best[0]:
            best = (med, th, len(ts))
    med, th, nt = best
    return {"value": round(n_lines / med, 4), "unit": "lines/s", "cores": th, "kind": "port",
            "sample": f"{n_lines} synthetic {height}x{width} fp32 lines, oracle forward+decode, 1 warm-up + {nt} timed (median), "
                      f"best of thread counts {cands} on a {ncpu}-cpu host"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="lines per GPU")
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-lines", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = try 16/32/64 threads, report the fastest")
    args = ap.parse_args()
    import faulthandler
    faulthandler.dump_traceback_later(420, exit=False, file=sys.stderr)      # diagnose hangs on the box

    from dtlr_amd import dist as ddist
    from dtlr_amd import ops, synth, weights
    from dtlr_amd.config import DTLRConfig
    from dtlr_amd.engine import DTLREngine
    from dtlr_amd.evaluation import decode_blank_records

    rank, local, world = ddist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    cfg = DTLRConfig.latin()
    sd = weights.synthetic_state_dict(cfg, seed=0)
    eng = DTLREngine(cfg, sd, dev, dtype)
    log(f"engine packed ({args.dtype}), rank {rank}/{world}")
    B = args.batch
    imgs = synth.noise_lines(B, args.height, args.width, seed=1000 + rank)     # this rank's shard
    x = torch.stack(imgs).to(dev)
    mask = torch.zeros((B, args.height, args.width), dtype=torch.bool, device=dev)
    n_total = B * world

    def local_step():
        out = eng.forward(x, mask, has_padding=False)
        return decode_blank_records(out)

    def step():
        labels, lengths = local_step()
        return ddist.all_gather_records(labels, lengths, n_total)

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    ops.MSDA_EVENTS = []
    ddist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec = step()
    torch.cuda.synchronize()
    ddist.barrier()
    elapsed = ddist.max_over_ranks(time.perf_counter() - t0, dev)
    events, ops.MSDA_EVENTS = ops.MSDA_EVENTS, None
    # MFMA-class launches are timed in a REPLAY of the same steps right after the timed region (same inputs, same stream):
    # ~170 event pairs per step inside it would cost ~2.5 ms of stream time per step and distort `value`.  Only launches of
    # >= 2 GFLOP are timed (ops.MFMA_EVENTS_MIN_FLOPS), so the event overhead stays below 2% of each measured launch.
    mfma_events = []
    if rank == 0:
        ops.MFMA_EVENTS = mfma_events
        for _ in range(min(args.steps, 5)):
            local_step()                                        # no collective here: the other ranks have left the timed region
        torch.cuda.synchronize()
        ops.MFMA_EVENTS = None
    replay_steps = min(args.steps, 5)
    log(f"timed {args.steps} steps in {elapsed:.3f}s")

    if rank != 0:
        ddist.finalize()                 # waits for rank 0 (replay + printing) at a last barrier, then tears the group down
        return
    S = 5440 * (args.height // 128) * (args.width // 2048) if (args.height, args.width) == (128, 2048) else None
    enc = [(a.elapsed_time(b), n, lq, s) for (a, b, n, lq, s) in events if lq == s]
    dec = [(a.elapsed_time(b), n, lq, s) for (a, b, n, lq, s) in events if lq != s]
    velem = 2 if dtype == torch.bfloat16 else 4
    roof = None
    if enc:
        ms = sum(e[0] for e in enc) / len(enc)
        n, lq, s = enc[0][1], enc[0][2], enc[0][3]
        alg = msda_algorithmic_bytes_per_line(s, lq, velem) * n
        achieved = alg / (ms * 1e-3) / 1e9
        traffic = None
        tj = os.path.join(ROOT, "profiles", "msda_traffic.json")
        if os.path.exists(tj):
            try:
                traffic = json.load(open(tj)).get(f"{args.dtype}_enc_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": "msda_enc_lds_kernel (deformable sampling, encoder call, Lq=S=5440/line)", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg, "mean_launch_ms": round(ms, 4), "launches_timed": len(enc)}
        if dec:
            msd = sum(e[0] for e in dec) / len(dec)
            n, lq, s = dec[0][1], dec[0][2], dec[0][3]
            algd = msda_algorithmic_bytes_per_line(s, lq, velem, ref_dim=4) * n
            roof["decoder_call"] = {"achieved": round(algd / (msd * 1e-3) / 1e9, 1), "mean_launch_ms": round(msd, 4)}
    # ---- per-class rooflines; `roofline` = the class with the largest share of the timed region ----
    by_kernel = []
    if roof:
        r = dict(roof)
        r["ms_per_step"] = round(sum(e[0] for e in enc) / args.steps, 3)
        by_kernel.append(r)
    classes = {}
    for (a, b, kind, flops, nbytes) in mfma_events:
        c = classes.setdefault(kind, [0.0, 0.0, 0, 0.0])
        c[0] += a.elapsed_time(b); c[1] += flops; c[2] += 1; c[3] += nbytes
    names = {"gemm_bf16": "gemm_ws_kernel<bf16> (every Linear / 1x1 conv / implicit-GEMM 3x3 conv, fused epilogues)",
             "gemm_f32": "gemm_ws_kernel<f32> (fp32 heads and selection scores, exact-fp32 MFMA 16x16x4)",
             "ffn_fused_bf16": "ffn_fused_bf16_kernel (linear1+ReLU+linear2+residual+LayerNorm, intermediate on chip)"}
    names["proj_ln_bf16"] = "proj_ln_bf16_kernel (attention output projection + residual + LayerNorm)"
    for kind, (ms, flops, cnt, nbytes) in classes.items():
        peak = MFMA_PEAK_F32_TFLOPS if kind == "gemm_f32" else MFMA_PEAK_BF16_TFLOPS
        ach = flops / (ms * 1e-3) / 1e12
        gbps = nbytes / (ms * 1e-3) / 1e9                        # compulsory operand + result bytes (each tensor once)
        # the binding roof of the class: most GEMMs of this path have K = 256 (or 64..128 in the first ResNet stage) and are
        # HBM-bound; both fractions are reported, `bound`/`achieved`/`peak`/`frac` name the larger one
        mf, hf = ach / peak, gbps / HBM_PEAK_GBS
        if hf > mf:
            head = {"bound": "hbm", "kernel": names.get(kind, kind), "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hf, 4)}
        else:
            head = {"bound": "mfma", "kernel": names.get(kind, kind), "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(mf, 4)}
        by_kernel.append({**head, "traffic": None, "mfma_tflops": round(ach, 1), "mfma_frac": round(mf, 4),
                          "hbm_gbps_algorithmic": round(gbps, 1), "hbm_frac": round(hf, 4),
                          "algorithmic_flops_per_launch": round(flops / cnt), "algorithmic_bytes_per_launch": round(nbytes / cnt),
                          "mean_launch_ms": round(ms / cnt, 4), "launches_timed": cnt, "ms_per_step": round(ms / replay_steps, 3),
                          "timed_in": "replay of the timed steps, launches >= 2 GFLOP only"})
    by_kernel.sort(key=lambda r: -r["ms_per_step"])
    dominant = by_kernel[0] if by_kernel else None
    line = {
        "metric": "text-lines/sec (128x2048, bs=32)", "value": round(n_total * args.steps / elapsed, 2), "unit": "lines/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"Latin DTLR (ResNet-50 + 6/6 deformable DETR, C=166) forward+decode, "
                               f"{B} synthetic {args.height}x{args.width} lines per GPU, random-init name-seeded weights",
                   "global_batch": n_total, "parallelism": f"dp{world}",
                   "library_backed_ops": sorted(ops.LIBRARY_BACKED)},
        "roofline": dominant,
        "roofline_by_kernel": by_kernel,
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.cpu_lines, args.height, args.width, repeats=2, threads=args.cpu_threads)
        line["speedup_vs_cpu"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
    print(json.dumps(line), flush=True)
    ddist.finalize()


if __name__ == "__main__":
    main()
