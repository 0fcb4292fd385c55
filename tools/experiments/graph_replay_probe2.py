#!/usr/bin/env python3
"""Round 5 follow-up of graph_replay_probe.py: WHERE does a HIP-graph replay of the forward stop reproducing itself after an intervening
eager forward?  The captured step returns the engine's debug tensors (backbone maps, tokens, encoder memory, two-stage scores / indices,
logits, boxes); each is cloned after replay 1, after replay 2, then again after an eager forward + replay 3, and compared stage by stage.
Also: the same sequence with the eager forward on ANOTHER input shape (workspace growth), and with only an eager torch op in between."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import synth, weights  # noqa: E402
from dtlr_amd.config import DTLRConfig  # noqa: E402
from dtlr_amd.engine import DTLREngine  # noqa: E402

dt = {"bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
cfg = DTLRConfig.latin()
eng = DTLREngine(cfg, weights.synthetic_state_dict(cfg, 0), "cuda:0", dt)
lines = synth.noise_lines(3, 128, 1024, seed=77)
mask = torch.zeros((1, 128, 1024), dtype=torch.bool, device="cuda:0")
x0, x1 = lines[0][None].cuda(), lines[1][None].cuda()
sx = x0.clone()


def step():
    out = eng.forward(sx, mask, has_padding=False, return_debug=True)
    d = out["_debug"]
    return {"feat0": d["feats"][0], "feat2": d["feats"][2], "src": d["src"], "memory": d["memory"], "topk_scores": d["topk_scores"],
            "topk_idx": d["topk_idx"], "logits": out["pred_logits"], "boxes": out["pred_boxes"]}


def snap(r):
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in r.items()}


def diff(a, b, tag):
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    print(f"{tag}: " + ("identical" if not bad else "DIFFER at " + ", ".join(f"{k} (max |d| {(a[k].float() - b[k].float()).abs().max().item():.3e})" for k in bad)), flush=True)
    return bad


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
e0 = snap(step())
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    res = step()
graph.replay(); g1 = snap(res)
graph.replay(); g2 = snap(res)
diff(e0, g1, "eager vs replay 1")
diff(g1, g2, "replay 1 vs replay 2")
torch.zeros(1 << 20, device="cuda:0").sum().item()
graph.replay(); g3 = snap(res)
diff(g1, g3, "replay 1 vs replay 3 (after an eager torch op)")
eng.forward(sx, mask, has_padding=False)
torch.cuda.synchronize()
graph.replay(); g4 = snap(res)
diff(g1, g4, "replay 1 vs replay 4 (after an eager forward, same input)")
eng.forward(x1, mask, has_padding=False)
torch.cuda.synchronize()
graph.replay(); g5 = snap(res)
diff(g1, g5, "replay 1 vs replay 5 (after an eager forward, other input)")
e5 = snap(step())
diff(e0, e5, "eager before vs eager after everything")
sx.copy_(x1); graph.replay(); g6 = snap(res)
e6 = snap(step())
diff(e6, g6, "eager(x1) vs replay(x1)")
