#!/usr/bin/env python3
"""Per-stage error budget of the bf16 (bench) engine against the fp32 engine and the CPU oracle.

    python tools/error_budget.py [--lines 8] [--config latin|chinese] [--oracle-lines 2] [--out profiles/r03_error_budget_vN.json]

(1) stage outputs of the bf16 engine vs the fp32 engine on the first `lines` lines of the BENCH batch, selection pinned to the fp32
    engine's: backbone maps, token matrix `src`, encoder `memory`, two-stage scores / reference points, per-decoder-layer `tgt` and
    `ref`, logits, boxes (max |d|, mean |d|, relative to the rms of the fp32 tensor);
(2) HYBRID forwards: bf16 up to a hand-over point, fp32 after it (and the reverse) -> which stage's rounding reaches the logits;
(3) the fp32 engine itself against O.dino_forward on `oracle-lines` lines (the fp32 engine is the yardstick of (1) and (2));
(4) decoded strings: CER of every variant against the fp32 engine's strings over ALL queries.
Test infrastructure: imports the oracle as the checker only."""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def stats(a, b):
    a, b = a.float(), b.float()
    d = (a - b).abs()
    rms = b.pow(2).mean().sqrt().item()
    return {"max": round(d.max().item(), 6), "mean": round(d.mean().item(), 7), "ref_rms": round(rms, 5),
            "rel_rms": round((d.pow(2).mean().sqrt().item()) / max(rms, 1e-12), 6)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=8)
    ap.add_argument("--config", default="latin")
    ap.add_argument("--oracle-lines", type=int, default=2)
    ap.add_argument("--strokes", action="store_true", help="stroke lines instead of the bench's noise lines")
    ap.add_argument("--out", default=None)
    ap.add_argument("--weights", type=int, default=None, help="generator version (default: weights.GENERATOR_VERSION; 3 = the undamped stress set)")
    ap.add_argument("--half", default="bf16", choices=["bf16", "f16"], help="the 16-bit engine under test")
    args = ap.parse_args()
    from dtlr_amd import synth, weights
    from dtlr_amd.config import DTLRConfig
    from dtlr_amd.engine import DTLREngine
    from dtlr_amd.evaluation import decode_blank
    from oracle import dtlr_oracle as O

    dev = torch.device("cuda:0")
    chinese = args.config == "chinese"
    cfg = DTLRConfig.chinese() if chinese else DTLRConfig.latin()
    sd = weights.synthetic_state_dict(cfg, seed=0, version=args.weights or weights.GENERATOR_VERSION)
    e32 = DTLREngine(cfg, sd, dev, torch.float32)
    e16 = DTLREngine(cfg, sd, dev, torch.bfloat16 if args.half == 'bf16' else torch.float16)
    n = args.lines
    W = 2560 if chinese else 2048
    if chinese:
        widths = synth.mixed_widths(32, [W - 1024, W - 768, W - 512, W - 256, W], seed=7)[:n]
        widths[0] = W
    else:
        widths = [W] * n
    lines = synth.stroke_lines(n, 128, widths, seed=31) if args.strokes else synth.noise_lines(n, 128, widths, seed=1000)
    x = torch.zeros((n, 3, 128, W))
    mask = torch.ones((n, 128, W), dtype=torch.bool)
    for i, im in enumerate(lines):
        x[i, :, :, :im.shape[2]] = im
        mask[i, :, :im.shape[2]] = False
    x, mask = x.to(dev), mask.to(dev)
    padded = chinese

    def run(eng_a, eng_b, handover, forced=None):
        """stages up to and including `handover` on eng_a, the rest on eng_b.  handover in
        {'none', 'backbone', 'tokens', 'enc0'..'enc5', 'encoder', 'two_stage', 'all'}"""
        order = ["none", "backbone", "tokens", "encoder", "two_stage", "all"]
        hi = order.index(handover)
        cap = {}
        eng = lambda k: eng_a if order.index(k) <= hi else eng_b
        E = eng("backbone")
        feats, last, level_hw = E.features(x)
        cap["feats"], cap["last"] = feats, last
        E2 = eng("tokens")
        feats = [f.to(E2.dtype) for f in feats]
        last = last.to(E2.dtype)
        src = E2.tokens(feats, last, level_hw)
        cap["src"] = src
        E3 = eng("encoder")
        g3 = E3.geometry_for(x, mask, level_hw, padded)
        memory = E3.encoder(src.to(E3.dtype), g3)
        cap["memory"] = memory
        E4 = eng("two_stage")
        g4 = E4.geometry_for(x, mask, level_hw, padded)
        ts = E4.two_stage(memory.to(E4.dtype), g4, forced)
        cap["scores"], cap["ref_unsig"], cap["topk_idx"] = ts["topk_scores"], ts["ref_unsig"], ts["topk_idx"]
        E5 = eng("all")
        g5 = E5.geometry_for(x, mask, level_hw, padded)
        if E5 is not E4:                                     # re-run the (cheap) two-stage gathers in the decoder engine's format
            ts = E5.two_stage(memory.to(E5.dtype), g5, ts["topk_idx"])
        dbg = {}
        hs, refs = E5.decoder(memory.to(E5.dtype), ts, g5, dbg=dbg)
        out = E5.heads(hs, refs, ts)
        cap["tgt"], cap["refs"] = dbg["tgt"], refs
        cap["pred_logits"], cap["pred_boxes"] = out["pred_logits"], out["pred_boxes"]
        return cap

    rep = {"half": args.half, "config": args.config, "lines": n, "input": "stroke" if args.strokes else "noise (bench batch)",
           "generator_version": args.weights or weights.GENERATOR_VERSION}
    with torch.no_grad():
        ref = run(e32, e32, "all")
        forced = ref["topk_idx"]
        ref_dec = decode_blank({"pred_logits": ref["pred_logits"], "pred_boxes": ref["pred_boxes"]})

        def cer(cap):
            got = decode_blank({"pred_logits": cap["pred_logits"].float(), "pred_boxes": cap["pred_boxes"].float()})
            dist = sum(O.levenshtein(a, b) for a, b in zip(ref_dec, got))
            return {"edits": dist, "chars": sum(len(a) for a in ref_dec)}

        def summary(cap):
            return {"logits": stats(cap["pred_logits"], ref["pred_logits"]), "boxes": stats(cap["pred_boxes"], ref["pred_boxes"]),
                    "cx_max": round((cap["pred_boxes"][..., 0].float() - ref["pred_boxes"][..., 0]).abs().max().item(), 6),
                    "cer_all_queries": cer(cap)}

        b = run(e16, e16, "all", forced)
        st = {}
        for i, (fa, fb) in enumerate(zip(b["feats"], ref["feats"])):
            st[f"backbone.layer{i + 2}"] = stats(fa, fb)
        st["input_proj.last"] = stats(b["last"], ref["last"])
        st["src"] = stats(b["src"], ref["src"])
        st["memory"] = stats(b["memory"], ref["memory"])
        st["two_stage.scores"] = stats(b["scores"], ref["scores"])
        st["two_stage.ref_unsig"] = stats(b["ref_unsig"], ref["ref_unsig"])
        for i in range(cfg.dec_layers):
            st[f"dec{i}.tgt"] = stats(b["tgt"][i], ref["tgt"][i])
            st[f"dec{i}.ref"] = stats(b["refs"][i + 1], ref["refs"][i + 1])
        rep["stages_bf16_vs_fp32"] = st
        rep["bf16_all"] = summary(b)
        hyb = {}
        for h in ("backbone", "tokens", "encoder", "two_stage"):
            hyb[f"bf16_through_{h}_then_fp32"] = summary(run(e16, e32, h, forced))
        for h in ("backbone", "tokens", "encoder", "two_stage"):
            hyb[f"fp32_through_{h}_then_bf16"] = summary(run(e32, e16, h, forced))
        rep["hybrids"] = hyb
        # free-running bf16 selection vs the fp32 one
        fr = run(e16, e16, "all")
        rep["free_selection_equal_frac"] = round((fr["topk_idx"] == forced).float().mean().item(), 4)
    if args.oracle_lines:
        k = args.oracle_lines
        torch.set_num_threads(min(16, os.cpu_count() or 8))
        o = O.dino_forward(sd, cfg, x[:k].cpu(), mask=mask[:k].cpu(), forced_topk=forced[:k].cpu())
        rep["fp32_engine_vs_oracle"] = {"logits_max": round((ref["pred_logits"][:k].cpu() - o["pred_logits"]).abs().max().item(), 7),
                                        "boxes_max": round((ref["pred_boxes"][:k].cpu() - o["pred_boxes"]).abs().max().item(), 8)}
    txt = json.dumps(rep, indent=1)
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)), exist_ok=True)
        open(os.path.join(ROOT, args.out), "w").write(txt + "\n")


if __name__ == "__main__":
    main()
