#!/usr/bin/env python3
"""Experiment: capture one bf16 forward + decode in a HIP graph (torch.cuda.graph) and compare replay time with eager launches."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dtlr_amd import synth, weights
from dtlr_amd.config import DTLRConfig
from dtlr_amd.engine import DTLREngine
from dtlr_amd.evaluation import decode_blank_records

dev = torch.device("cuda:0")
cfg = DTLRConfig.latin()
eng = DTLREngine(cfg, weights.synthetic_state_dict(cfg, seed=0), dev, torch.bfloat16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.stack(synth.noise_lines(B, 128, 2048, seed=1000)).to(dev)
mask = torch.zeros((B, 128, 2048), dtype=torch.bool, device=dev)

def step():
    out = eng.forward(x, mask, has_padding=False)
    lab, ln = decode_blank_records(out)
    return out["pred_logits"], lab, ln

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3

eager = timeit(step)
ref = [t.clone() for t in step()]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        outs = step()
    g.replay(); torch.cuda.synchronize()
    same = [bool(torch.equal(a, b)) for a, b in zip(outs, ref)]
    graphed = timeit(g.replay)
    print(json.dumps({"eager_ms": eager, "graph_ms": graphed, "outputs_equal": same}))
except Exception as e:
    print(json.dumps({"eager_ms": eager, "graph_error": repr(e)[:500]}))
