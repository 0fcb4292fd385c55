#!/usr/bin/env python3
"""Per-operator timeline of one bench step: every dtlr_amd.ops call is bracketed with HIP events and aggregated by
(operator, argument shapes).  python tools/profile_ops.py [--steps 3] [--top 60]
Event pairs add a few microseconds to small launches: use for ranking, not for absolute numbers."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32s", "f32"])
    args = ap.parse_args()
    from dtlr_amd import ops, synth, weights
    from dtlr_amd.config import DTLRConfig
    from dtlr_amd.engine import DTLREngine
    from dtlr_amd.evaluation import decode_blank_records
    dev = torch.device("cuda:0")
    cfg = DTLRConfig.latin()
    eng = DTLREngine(cfg, weights.synthetic_state_dict(cfg, seed=0), dev,
                     {"bf16": torch.bfloat16, "f16": torch.float16}.get(args.dtype, torch.float32), split=args.dtype == "f32s")
    x = torch.stack(synth.noise_lines(args.batch, 128, 2048, seed=1000)).to(dev)
    mask = torch.zeros((args.batch, 128, 2048), dtype=torch.bool, device=dev)
    spans = []
    names = ["linear", "layernorm", "conv2d_nhwc", "maxpool_nhwc", "groupnorm_tokens", "msda", "msda_fused", "msda_encoder", "mha",
             "ffn_fused", "ffn32", "ffn4", "proj_ln", "proj_ln_k256", "proj_ln_split", "gemm_k256", "gemm_kres", "gemm_kres_chain", "dec_query_stage", "stem_conv7x7_pool", "linear_rowmax", "geometry", "two_stage_gather", "stem_conv7x7", "stem_conv7x7_f32", "linear_resbcast", "box_mlp_refine", "blank_emissions", "box_head_refine", "decoder_query_prep", "box_refine", "topk_rows", "decode_blank", "ffn_split", "gemm_k256s", "gemm_k256s_multi", "stem_conv7x7_f32s", "head_ts", "gemm_kres_bcast384", "gemm_kres_cat_s2"]

    def wrap(name):
        fn = getattr(ops, name)

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            shp = tuple(tuple(t.shape) for t in a[:2] if torch.is_tensor(t))
            dt = str(a[0].dtype).replace("torch.", "") if torch.is_tensor(a[0]) else ""
            if name == "gemm_k256s" and k.get("residual") is not None:
                name_ = name + "+res+ln"
            else:
                name_ = name
            extra = "+a2" if k.get("a2") is not None or (name == "linear" and len(a) > 5 and a[5] is not None) else ""
            spans.append((name_ + extra, dt, shp, e0, e1))
            return r
        setattr(ops, name, timed)

    for n in names:
        if hasattr(ops, n):
            wrap(n)

    def step():
        out = eng.forward(x, mask, has_padding=False)
        decode_blank_records(out)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    spans.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    agg = {}
    for name, dt, shp, a, b in spans:
        k = (name, dt, shp)
        v = agg.setdefault(k, [0, 0.0])
        v[0] += 1
        v[1] += a.elapsed_time(b)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values()) / args.steps
    print(json.dumps({"step_ms_with_events": round(e0.elapsed_time(e1) / args.steps, 3), "ops_ms": round(tot, 3)}))
    for (name, dt, shp), (cnt, ms) in rows[: args.top]:
        print(f"{ms / args.steps:8.3f} ms/step  x{cnt // args.steps:<3d} {ms / cnt * 1e3:8.1f} us  {name:18s} {dt:8s} {shp}")


if __name__ == "__main__":
    main()
