#!/usr/bin/env python3
"""Experiment: does running the bench batch as TWO concurrent half-batches (one HIP stream each) beat one 32-line pass?
Lines are independent, so the split changes no result; the question is whether kernels with different bottlenecks (MFMA-bound FFN,
LDS/VALU-bound MSDA, HBM-bound early convolutions, sub-round decoder grids) overlap on the chip.
    python tools/experiments/two_stream_probe.py [--steps 20] [--batch 32] [--parts 2]
Prints one JSON line: ms per 32-line step for 1 stream and for the split, plus the host's launch-issue time per step."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--parts", type=int, nargs="+", default=[2])
    args = ap.parse_args()
    from dtlr_amd import synth, weights
    from dtlr_amd.config import DTLRConfig
    from dtlr_amd.engine import DTLREngine
    from dtlr_amd.evaluation import decode_blank_records
    dev = torch.device("cuda:0")
    cfg = DTLRConfig.latin()
    eng = DTLREngine(cfg, weights.synthetic_state_dict(cfg, seed=0), dev, torch.bfloat16)
    B = args.batch
    x = torch.stack(synth.noise_lines(B, 128, 2048, seed=1000)).to(dev)
    mask = torch.zeros((B, 128, 2048), dtype=torch.bool, device=dev)

    def one(xs, ms):
        return decode_blank_records(eng.forward(xs, ms, has_padding=False))

    def timed(fn, steps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        issue = 0.0
        for _ in range(steps):
            ti = time.perf_counter()
            fn()
            issue += time.perf_counter() - ti
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, issue / steps * 1e3

    res = {}
    ref = one(x, mask)
    res["one_stream"] = dict(zip(("ms_per_step", "host_issue_ms"), timed(lambda: one(x, mask), args.steps)))
    for parts in args.parts:
        streams = [torch.cuda.Stream() for _ in range(parts)]
        per = B // parts
        xs = [x[i * per:(i + 1) * per].contiguous() for i in range(parts)]
        mk = [mask[i * per:(i + 1) * per].contiguous() for i in range(parts)]
        outs = [None] * parts

        def split():
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    outs[i] = one(xs[i], mk[i])
        torch.cuda.synchronize()
        ms, issue = timed(split, args.steps)
        torch.cuda.synchronize()
        same = all(torch.equal(torch.cat([o[k] for o in outs]), ref[k]) for k in range(len(ref))) if isinstance(ref, (tuple, list)) else None
        res[f"split_{parts}"] = {"ms_per_step": ms, "host_issue_ms": issue, "records_equal_one_stream": same}
        # the same half-batches back to back on ONE stream: separates "smaller batches" from "overlap"
        def serial():
            for i in range(parts):
                outs[i] = one(xs[i], mk[i])
        ms, issue = timed(serial, args.steps)
        res[f"serial_{parts}"] = {"ms_per_step": ms, "host_issue_ms": issue}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
