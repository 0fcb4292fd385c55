#!/usr/bin/env python3
"""Stage split of one bench step (HIP events around the engine's stage methods, averaged).
    python tools/profile_stages.py [--steps 5] [--dtype bf16|f16|f32s|f32] [--batch 32]
Prints one JSON line: ms per stage.  Same workload as bench.py (Latin, 128x2048, synthetic)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--convs", action="store_true", help="also time every backbone convolution (name, shape, ms, GB/s, TFLOP/s)")
    ap.add_argument("--engine-opt", action="append", default=[], metavar="NAME=VALUE", help="set a DTLREngine attribute (as bench.py --engine-opt)")
    args = ap.parse_args()
    from dtlr_amd import synth, weights
    from dtlr_amd.config import DTLRConfig
    from dtlr_amd.engine import DTLREngine
    from dtlr_amd.evaluation import decode_blank_records
    dev = torch.device("cuda:0")
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32s": torch.float32}[args.dtype]
    cfg = DTLRConfig.latin()
    eng = DTLREngine(cfg, weights.synthetic_state_dict(cfg, seed=0), dev, dtype, split=args.dtype == "f32s")
    for kv in args.engine_opt:
        k, _, v = kv.partition("=")
        setattr(eng, k, type(getattr(eng, k))(int(v)) if isinstance(getattr(eng, k), (bool, int)) else float(v))
    x = torch.stack(synth.noise_lines(args.batch, 128, 2048, seed=1000)).to(dev)
    mask = torch.zeros((args.batch, 128, 2048), dtype=torch.bool, device=dev)
    spans = {}

    def wrap(obj, name, label=None):
        fn = getattr(obj, name)
        label = label or name

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            spans.setdefault(label, []).append((e0, e1))
            return r
        setattr(obj, name, timed)

    for n in ("backbone", "encoder", "two_stage", "decoder"):
        wrap(eng, n)
    conv_info = {}
    if args.convs:
        orig_conv = eng._conv

        def timed_conv(name, x, stride, padding, relu=False, residual=None):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = orig_conv(name, x, stride, padding, relu, residual)
            e1.record()
            w = eng.w[name + ".w"]
            k = w.shape[1] * (w.shape[2] if w.dim() == 4 else 1) * (w.shape[3] if w.dim() == 4 else 1) if w.dim() == 4 else w.shape[1]
            flops = 2.0 * y.numel() * k
            byts = (x.numel() / (stride * stride if w.dim() == 2 else 1) + y.numel() * (2 if residual is not None else 1)) * x.element_size()
            conv_info[name] = (tuple(x.shape), tuple(y.shape), flops, byts)
            spans.setdefault("conv:" + name, []).append((e0, e1))
            return y
        eng._conv = timed_conv

    def step():
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        out = eng.forward(x, mask, has_padding=False)
        e1.record()
        decode_blank_records(out)
        e2.record()
        spans.setdefault("forward_total", []).append((e0, e1))
        spans.setdefault("decode_blank", []).append((e1, e2))

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    spans.clear()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    res = {k: round(sum(a.elapsed_time(b) for a, b in v) / args.steps, 3) for k, v in spans.items() if not k.startswith("conv:")}
    for k, v in spans.items():
        if k.startswith("conv:"):
            ms = sum(a.elapsed_time(b) for a, b in v) / args.steps
            xin, yout, flops, byts = conv_info[k[5:]]
            print(json.dumps({"conv": k[5:], "in": xin, "out": yout, "ms": round(ms, 4), "TFLOPs": round(flops / ms / 1e9, 1), "GBps_min": round(byts / ms / 1e6, 1)}))
    res["input_proj_heads_other"] = round(res["forward_total"] - sum(res[k] for k in ("backbone", "encoder", "two_stage", "decoder")), 3)
    print(json.dumps({"stage_ms": res, "dtype": args.dtype, "batch": args.batch}))


if __name__ == "__main__":
    main()
