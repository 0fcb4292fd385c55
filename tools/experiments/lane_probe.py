#!/usr/bin/env python3
"""Round 4: the "lanes 48..63" effect as a victim / aggressor experiment on ONE GPU.
  victim   : a loop of ONE operator on fixed inputs; every result is compared with the first ON THE DEVICE (no synchronisation in the loop);
             reports how many calls differed and the histogram of differing 8-channel pieces.
  aggressor: a second process that runs one kind of work in a loop for a fixed time.
      python tools/experiments/lane_probe.py victim <op> <calls> [tag]          op: msda_fused_f32 | msda_fused_bf16 | msda_enc_f32 | gemm_f32s | mha_f32s
      python tools/experiments/lane_probe.py aggressor <kind> <seconds>         kind: none | fwd_f32s | fwd_f32 | fwd_bf16 | gemm_f32s | gemm_f32 | mha_f32s | mha_f32 | msda_enc_f32 | msda_fused_f32 | ln_f32 | copy"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import ops, synth, weights  # noqa: E402
from dtlr_amd.config import DTLRConfig  # noqa: E402
from dtlr_amd.engine import DTLREngine  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
LHW = [(16, 256), (8, 128), (4, 64), (2, 32)]
S = sum(h * w for h, w in LHW)


def rnd(*shape, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(dev)


def make_op(kind, B=8):
    shapes = torch.as_tensor(LHW, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    if kind in ("msda_fused_f32", "msda_fused_bf16"):
        dt = torch.float32 if kind.endswith("f32") else torch.bfloat16
        vall = rnd(B, S, 1536).to(dt)
        value = vall[..., 256:512].unflatten(-1, (8, 32))          # a column slice, as in the decoder
        ow = rnd(B, 900, 384)
        ow[..., :256] *= 2.0
        ow = ow.to(dt)
        ref = torch.rand((B, 900, 4, 4), generator=g).to(dev) * torch.tensor([1.0, 1.0, 0.1, 0.3], device=dev)
        return lambda: ops.msda_fused(value, shapes, lsi, ow, ref)
    if kind == "msda_op_f32":                  # the B1 operator's hot kernel, msda_fwd_l4p4_kernel<float, 4>
        value = rnd(B, S, 8, 32)
        loc = torch.rand((B, 900, 8, 4, 4, 2), generator=g).to(dev)
        aw = torch.softmax(rnd(B, 900, 8, 16), -1).reshape(B, 900, 8, 4, 4).contiguous()
        return lambda: ops.msda(value, shapes, lsi, loc, aw)
    if kind in ("gemm_bf16", "gemm_bf16_f32out", "gemm_f16"):
        dt = torch.float16 if kind == "gemm_f16" else torch.bfloat16
        x, w, b = rnd(B * S, 256).to(dt), rnd(2048, 256, scale=1 / 16.0).to(dt), rnd(2048)
        od = torch.float32 if kind.endswith("f32out") else dt
        return lambda: ops.linear(x, w, b, relu=True, out_dtype=od)
    if kind == "mha_bf16":
        qk, v = rnd(B, 900, 512).bfloat16(), rnd(B, 900, 256).bfloat16()
        return lambda: ops.mha(qk, v, 8)
    if kind == "ffn_bf16":
        x = rnd(B * S, 256).bfloat16()
        w1, w2 = rnd(2048, 256, scale=1 / 16.0).bfloat16(), rnd(256, 2048, scale=1 / 45.0).bfloat16()
        w1p, w2p = ops.ffn32_pack(w1, w2)
        b1, b2, lw, lb = rnd(2048), rnd(256), rnd(256), rnd(256)
        return lambda: ops.ffn32(x, w1p, b1, w2p, b2, lw, lb)
    if kind == "msda_enc_bf16":
        value = rnd(B, S, 8, 32).bfloat16()
        ow = rnd(B, S, 384)
        ow[..., :256] *= 2.0
        ow = ow.bfloat16()
        rp = torch.cat([torch.stack(torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h, torch.linspace(0.5, w - 0.5, w) / w, indexing="ij")[::-1], -1).reshape(-1, 2)
                        for h, w in LHW], 0)
        ref = rp[None, :, None, :].expand(B, S, 4, 2).contiguous().to(dev)
        return lambda: ops.msda_encoder(value, LHW, ow, ref)
    if kind == "topk":
        sc = rnd(B * 4, S)
        return lambda: ops.topk_rows(sc, 900)
    if kind == "msda_enc_f32":
        value = rnd(B, S, 8, 32)
        ow = rnd(B, S, 384)
        ow[..., :256] *= 2.0
        rp = torch.cat([torch.stack(torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h, torch.linspace(0.5, w - 0.5, w) / w, indexing="ij")[::-1], -1).reshape(-1, 2)
                        for h, w in LHW], 0)
        ref = rp[None, :, None, :].expand(B, S, 4, 2).contiguous().to(dev)
        return lambda: ops.msda_encoder(value, LHW, ow, ref)
    if kind in ("gemm_f32s", "gemm_f32"):
        x, w = rnd(B * S, 256), rnd(2048, 256, scale=1 / 16.0)
        wk = ops.split_pack(w) if kind == "gemm_f32s" else w
        b = rnd(2048)
        return lambda: ops.linear(x, wk, b, relu=True)
    if kind in ("mha_f32s", "mha_f32"):
        qk, v = rnd(B, 900, 512), rnd(B, 900, 256)
        return lambda: ops.mha(qk, v, 8, split=kind == "mha_f32s")
    if kind == "ln_f32":
        x, r, w, b = rnd(B * S, 256), rnd(B * S, 256), rnd(256), rnd(256)
        return lambda: ops.layernorm(x, w, b, 1e-5, r)
    if kind == "copy":
        x = rnd(1 << 26)
        return lambda: x.clone()
    if kind.startswith("fwd_"):
        k = kind[4:]
        cfg = DTLRConfig.latin()
        eng = DTLREngine(cfg, weights.synthetic_state_dict(cfg, seed=0), dev, {"bf16": torch.bfloat16, "f16": torch.float16}.get(k, torch.float32), split=k == "f32s")
        x = torch.stack(synth.noise_lines(3, 128, 2048, seed=1000)).to(dev)
        mask = torch.zeros((3, 128, 2048), dtype=torch.bool, device=dev)
        return lambda: eng.forward(x, mask, has_padding=False)["pred_logits"]
    raise SystemExit(f"unknown kind {kind}")


def main():
    role, kind = sys.argv[1], sys.argv[2]
    if role == "aggressor":
        secs = float(sys.argv[3])
        if kind == "none":
            time.sleep(secs)
            return
        op = make_op(kind)
        op()
        torch.cuda.synchronize()
        t0, n = time.time(), 0
        while time.time() - t0 < secs:
            for _ in range(20):
                op()
            torch.cuda.synchronize()
            n += 20
        print(f"aggressor {kind}: {n} calls in {secs:.0f}s", flush=True)
        return
    calls = int(sys.argv[3])
    tag = sys.argv[4] if len(sys.argv) > 4 else ""
    op = make_op(kind)
    ref = op().clone()
    torch.cuda.synchronize()
    C = ref.shape[-1]
    bad_calls = torch.zeros((), dtype=torch.int64, device=dev)
    hist = torch.zeros(C // 8, dtype=torch.int64, device=dev)
    t0 = time.time()
    for i in range(calls):
        d = op() != ref
        bad_calls += d.any()
        hist += d.reshape(-1, C // 8, 8).any(-1).sum(0)
        if i % 50 == 49:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(json.dumps({"victim": kind, "tag": tag, "calls": calls, "bad_calls": int(bad_calls), "bad_piece8_hist": hist.tolist() if int(bad_calls) else None,
                      "seconds": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
