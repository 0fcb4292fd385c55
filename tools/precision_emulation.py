#!/usr/bin/env python3
"""Precision design study: the fp32 HIP engine with ROUNDINGS INJECTED where a 16-bit engine stores or feeds
16-bit values, per stage, so that a storage format can be priced before any kernel is written.

    python tools/precision_emulation.py [--lines 8] [--out profiles/r03_precision_emulation_vN.json]

A variant is `backbone,tokens,encoder,decoder` with each stage one of
    f32          no rounding
    bf16 | f16   every operator output is rounded to that format (the storage-rounded engine), GEMM weights rounded to it
    bf16s | f16s the same operands, but the residual stream / LayerNorm outputs stay fp32 (rounded only as GEMM A-operands)
`bf16,bf16,bf16,bf16` must reproduce the error level of the real bf16 engine (printed beside it): that validates the emulation.
Reports logit / box / cx error and the all-queries CER against the un-rounded fp32 engine, selection pinned."""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dtlr_amd import ops, synth, weights  # noqa: E402
from dtlr_amd.config import DTLRConfig  # noqa: E402
from dtlr_amd.engine import DTLREngine  # noqa: E402

H = {"bf16": torch.bfloat16, "f16": torch.float16}


class Emul(DTLREngine):
    """fp32 engine; self.cur = (half dtype or None, stream32) is set per stage by the driver."""

    def __init__(self, cfg, sd, dev):
        super().__init__(cfg, sd, dev, torch.float32)
        self.cur = (None, False)
        self._wr = {}

    def R(self, t):
        h = self.cur[0]
        return t if h is None else t.to(h).float()

    def RS(self, t):                      # a residual-stream / LayerNorm output
        return t if self.cur[1] else self.R(t)

    def W(self, name):
        h = self.cur[0]
        if h is None:
            return self.w[name]
        key = (name, h)
        if key not in self._wr:
            self._wr[key] = self.w[name].to(h).float()
        return self._wr[key]

    def _lin(self, name, x, relu=False, residual=None, a2=None, row_mask=None, out_dtype=None, round_out=True):
        xin = x if a2 is None else x + a2
        y = ops.linear(self.R(xin).contiguous(), self.W(name + ".w"), self.w[name + ".b"], relu, residual, None, row_mask, None)
        return self.R(y) if round_out else y

    def _conv(self, name, x, stride, padding, relu=False, residual=None):
        w = self.W(name + ".w")
        if w.dim() == 2:
            if stride != 1:
                y = ops.conv2d_nhwc(x, w.view(w.shape[0], 1, 1, w.shape[1]), self.w[name + ".b"], stride, 0, relu, residual)
            else:
                y = ops.linear(x, w, self.w[name + ".b"], relu=(2 if relu else 0), residual=residual)
        else:
            y = ops.conv2d_nhwc(x, w, self.w[name + ".b"], stride, padding, relu, residual)
        return self.R(y)

    def _proj_ln(self, proj, norm, a, residual):
        y = self._lin(proj, a, round_out=False)
        return self.RS(ops.layernorm(y, self.w[norm + ".w"], self.w[norm + ".b"], 1e-5, residual))

    def _ffn(self, q, norm, x):
        h = self._lin(q + "ff1", x, relu=True)
        y = self._lin(q + "ff2", h, round_out=False)
        return self.RS(ops.layernorm(y, self.w[q + norm + ".w"], self.w[q + norm + ".b"], 1e-5, x))

    def _ln(self, name, x, residual=None):
        return self.RS(ops.layernorm(x, self.w[name + ".w"], self.w[name + ".b"], 1e-5, residual))

    def backbone(self, x_nchw):
        x = ops.stem_conv7x7_f32(self.R(x_nchw), self.W("conv1.wk"))
        x = self.R(ops.maxpool_nhwc(x, bias=self.w["conv1.b"], relu=True))
        outs = []
        for li, nblocks in enumerate(self.cfg.backbone_blocks, start=1):
            for bi in range(nblocks):
                q = f"l{li}.{bi}."
                stride = 2 if (bi == 0 and li > 1) else 1
                idt = self._conv(q + "ds", x, stride, 0) if bi == 0 else x
                o = self._conv(q + "c1", x, 1, 0, relu=True)
                o = self._conv(q + "c2", o, stride, 1, relu=True)
                x = self._conv(q + "c3", o, 1, 0, relu=True, residual=idt)
            if li >= 2:
                outs.append(x)
        return outs

    def tokens(self, feats, last, level_hw):
        return self.R(super().tokens(feats, last, level_hw))

    def _msda_module(self, *a, **k):
        return self.R(super()._msda_module(*a, **k))

    def _box_mlp_hidden(self, name, x):
        h = ops.linear(self.R(x.float()), self.W(name + "0.w"), self.w[name + "0.b"], relu=True)
        return ops.linear(self.R(h), self.W(name + "1.w"), self.w[name + "1.b"], relu=True)

    def decoder(self, memory, ts, g, want_aux=False, dbg=None):
        real_mha = ops.mha
        ops.mha = lambda qk, v, nh: self.R(real_mha(qk, v, nh))
        try:
            return super().decoder(memory, ts, g, want_aux, dbg)
        finally:
            ops.mha = real_mha


def parse(spec):
    out = []
    for s in spec.split(","):
        if s == "f32":
            out.append((None, False))
        else:
            out.append((H[s.rstrip("s")], s.endswith("s")))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=8)
    ap.add_argument("--strokes", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--variants", default=None, help="semicolon-separated variant specs")
    args = ap.parse_args()
    from dtlr_amd.evaluation import decode_blank
    from oracle import dtlr_oracle as O              # Levenshtein only (checker)
    dev = torch.device("cuda:0")
    cfg = DTLRConfig.latin()
    sd = weights.synthetic_state_dict(cfg, seed=0)
    n, Wd = args.lines, 2048
    lines = synth.stroke_lines(n, 128, Wd, seed=31) if args.strokes else synth.noise_lines(n, 128, Wd, seed=1000)
    x = torch.stack(lines).to(dev)
    mask = torch.zeros((n, 128, Wd), dtype=torch.bool, device=dev)
    E = Emul(cfg, sd, dev)

    def run(spec, forced=None):
        fb, ft, fe, fd = parse(spec)
        E.cur = fb
        feats, last, level_hw = E.features(x)
        E.cur = ft
        src = E.tokens(feats, last, level_hw)
        g = E.geometry_for(x, mask, level_hw, False)
        E.cur = fe
        memory = E.encoder(src, g)
        E.cur = (None, False)                        # two-stage: fp32-grade in every engine (split products)
        ts = E.two_stage(memory, g, forced)
        E.cur = fd
        hs, refs = E.decoder(memory, ts, g)
        out = E.heads(hs, refs, ts)
        E.cur = (None, False)
        return out["pred_logits"], out["pred_boxes"], ts["topk_idx"]

    variants = (args.variants.split(";") if args.variants else
                ["bf16,bf16,bf16,bf16", "f16,f16,f16,f16", "bf16,f16,f16,f16", "bf16,bf16,f16,f16", "bf16,bf16,bf16,f16",
                 "bf16,bf16,bf16,f16s", "bf16,bf16,bf16,bf16s", "bf16,bf16,bf16s,bf16s", "bf16,f16,f16s,f16s", "bf16,bf16,bf16,f32",
                 "bf16,f16,f16,f32", "f16,f16,f16s,f16s", "bf16,f32,f32,f32", "f16,f32,f32,f32"])
    rep = {"lines": n, "input": "stroke" if args.strokes else "noise (bench batch)", "generator_version": weights.GENERATOR_VERSION, "variants": {}}
    with torch.no_grad():
        rl, rb, forced = run("f32,f32,f32,f32")
        ref_dec = decode_blank({"pred_logits": rl, "pred_boxes": rb})
        chars = sum(len(a) for a in ref_dec)
        for v in variants:
            l, b, _ = run(v, forced)
            got = decode_blank({"pred_logits": l, "pred_boxes": b})
            dl = (l - rl).abs()
            rep["variants"][v] = {"logit_max": round(dl.max().item(), 5), "logit_mean": round(dl.mean().item(), 6),
                                  "box_max": round((b - rb).abs().max().item(), 6), "cx_max": round((b[..., 0] - rb[..., 0]).abs().max().item(), 6),
                                  "cx_mean": round((b[..., 0] - rb[..., 0]).abs().mean().item(), 7),
                                  "edits": sum(O.levenshtein(p, q) for p, q in zip(ref_dec, got)), "chars": chars}
            print(v, rep["variants"][v], flush=True)
    if args.out:
        open(os.path.join(ROOT, args.out), "w").write(json.dumps(rep, indent=1) + "\n")


if __name__ == "__main__":
    main()
