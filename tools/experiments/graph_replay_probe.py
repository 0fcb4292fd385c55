#!/usr/bin/env python3
"""Round 4 experiment (NOT adopted): forward + blank decode of ONE line captured in a HIP graph (torch.cuda.CUDAGraph) and replayed.
Findings on MI355X: (1) no latency gain -- 2.82 ms per replay against 2.83 ms for the eager step: at one line the ~200 dependent launches
are bound by GPU-side dispatch latency, not by the host; (2) a replay after an intervening EAGER forward of the same engine did not reproduce
the first replay's result (the first replay equals the eager result bit for bit).  The product keeps plain stream launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import synth, weights  # noqa: E402
from dtlr_amd.config import DTLRConfig  # noqa: E402
from dtlr_amd.engine import DTLREngine  # noqa: E402
from dtlr_amd.evaluation import decode_blank_records  # noqa: E402


def graphed_step(eng, x, mask):
    sx, sm = x.clone(), mask.clone()

    def step():
        out = eng.forward(sx, sm, has_padding=False)
        lab, ln = decode_blank_records(out)
        return lab, ln, out
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        res = step()

    def replay(x_new=None):
        if x_new is not None:
            sx.copy_(x_new)
        graph.replay()
        return res
    replay.graph = graph
    return replay


cfg = DTLRConfig.latin()
eng = DTLREngine(cfg, weights.synthetic_state_dict(cfg, 0), "cuda:0", torch.bfloat16)
lines = synth.noise_lines(3, 128, 1024, seed=77)
mask = torch.zeros((1, 128, 1024), dtype=torch.bool, device="cuda:0")
x0, x1 = lines[0][None].cuda(), lines[1][None].cuda()
e0 = eng.forward(x0, mask, has_padding=False)["pred_logits"].clone()
e1 = eng.forward(x1, mask, has_padding=False)["pred_logits"].clone()
rp = graphed_step(eng, x0, mask)
g0 = rp()[2]["pred_logits"].clone()
print("graph(x0) == eager(x0):", torch.equal(g0, e0))
g1 = rp(x1)[2]["pred_logits"].clone()
g1r = rp(x1)[2]["pred_logits"].clone()
print("graph(x1) == eager(x1):", torch.equal(g1, e1), " replay twice equal:", torch.equal(g1, g1r))
e1b = eng.forward(x1, mask, has_padding=False)["pred_logits"].clone()
g1b = rp(x1)[2]["pred_logits"].clone()
print("eager(x1) after capture == before:", torch.equal(e1b, e1), " graph(x1) after an eager forward == first graph(x1):", torch.equal(g1b, g1))
