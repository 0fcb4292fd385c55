#!/usr/bin/env python3
"""Round 4: which operator of the f32 / f32s engines differs between forwards (alone, or with a second process on the same GPU)?
Every dtlr_amd.ops call is hooked, its result check-summed ON THE DEVICE (no synchronisation inside a forward); after each forward the
checksums are compared with the first forward's; for the first differing operator: shape, differing elements / rows / 8-channel pieces,
whether its arguments equal the reference forward's, and what a re-run on the same (still alive) arguments gives.
    python tools/experiments/contention_probe2.py <forwards> <tag> <engine: f32s|f32|bf16|f16> [B]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import ops, synth, weights  # noqa: E402
from dtlr_amd.config import DTLRConfig  # noqa: E402
from dtlr_amd.engine import DTLREngine  # noqa: E402

N, tag, kind = int(sys.argv[1]), sys.argv[2], sys.argv[3]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
cfg = DTLRConfig.latin()
dt = {"bf16": torch.bfloat16, "f16": torch.float16}.get(kind, torch.float32)
eng = DTLREngine(cfg, weights.synthetic_state_dict(cfg, seed=0), dev, dt, split=kind == "f32s")
x = torch.stack(synth.noise_lines(B, 128, 2048, seed=1000)).to(dev)
mask = torch.zeros((B, 128, 2048), dtype=torch.bool, device=dev)
eng.forward(x, mask, has_padding=False)
log = []
names = [n for n in dir(ops) if callable(getattr(ops, n)) and not n.startswith("_") and n not in ("require_cuda", "msda_encoder_far_fraction", "msda_encoder_fits", "split_pack")
         and getattr(getattr(ops, n), "__module__", "") == "dtlr_amd.ops" and not isinstance(getattr(ops, n), type)]
real = {}


def csum(r):
    ts = [t for t in (r if isinstance(r, (tuple, list)) else (r,)) if torch.is_tensor(t) and t.is_cuda]
    acc = None
    for t in ts:
        v = t.detach().contiguous().view(torch.uint8)
        n4 = v.numel() // 4 * 4
        s = v[:n4].view(torch.int32).sum(dtype=torch.int64) + v[n4:].sum(dtype=torch.int64)
        acc = s if acc is None else acc * 31 + s
    return acc if acc is not None else torch.zeros((), dtype=torch.int64, device=dev)


for n in names:
    fn = getattr(ops, n)
    real[n] = fn

    def wrap(fn=fn, n=n):
        def w(*a, **k):
            r = fn(*a, **k)
            log.append((n, a, k, r, csum(r)))
            return r
        return w
    setattr(ops, n, wrap())


def first(r):
    return r if torch.is_tensor(r) else [t for t in r if torch.is_tensor(t)][0]


ref_log = None
found = {}
nrep = 0
t0 = time.time()
for i in range(N):
    log.clear()
    eng.forward(x, mask, has_padding=False)
    sums = torch.stack([c for (_, _, _, _, c) in log]).cpu()          # the only synchronisation of the forward
    if ref_log is None:
        ref_log, ref_sums = list(log), sums
        print(tag, "ops per forward:", len(ref_log), flush=True)
        continue
    neq = (sums != ref_sums).nonzero().flatten()
    if neq.numel() == 0:
        continue
    j = int(neq[0])
    n, args, kw, out, _ = log[j]
    shp = tuple(tuple(t.shape) for t in args[:2] if torch.is_tensor(t))
    found[f"{j}:{n}:{shp}"] = found.get(f"{j}:{n}:{shp}", 0) + 1
    if nrep >= 5:
        continue
    nrep += 1
    rep = {"forward": i, "op_index": j, "op": n, "arg_shapes": [list(s) for s in shp], "ops_differing_after": int(neq.numel())}
    o1, o0 = first(out), first(ref_log[j][3])
    d = (o1 != o0)
    rep["elements_differ"] = int(d.sum()); rep["shape"] = list(o1.shape)
    rep["max_abs_diff"] = float((o1.float() - o0.float()).abs().max())
    idx = d.nonzero()
    rows = torch.unique(idx[:, :-1], dim=0)
    rep["distinct_rows"] = int(rows.shape[0]); rep["first_rows"] = rows[:10].tolist()
    C = o1.shape[-1]
    if C % 8 == 0 and C <= 2048:
        rep["by_piece8"] = torch.bincount(idx[:, -1] // 8, minlength=C // 8).tolist()
    a_eq = []
    for u, v in zip(args, ref_log[j][1]):
        if torch.is_tensor(u) and torch.is_tensor(v) and u.shape == v.shape:
            a_eq.append(bool(torch.equal(u, v)))
    rep["args_equal_reference_args"] = a_eq
    again = first(real[n](*args, **kw))
    torch.cuda.synchronize()
    rep["rerun_equals_reference"] = bool(torch.equal(again, o0)); rep["rerun_equals_this"] = bool(torch.equal(again, o1))
    if rows.shape[0]:
        r0 = tuple(rows[0].tolist())
        c0 = int(idx[0, -1]) // 8 * 8
        rep["bad"] = [round(v, 6) for v in o1[r0].float()[c0:c0 + 8].tolist()]
        rep["good"] = [round(v, 6) for v in o0[r0].float()[c0:c0 + 8].tolist()]
    print(tag, json.dumps(rep), flush=True)
print(tag, "done", json.dumps({"forwards": N, "engine": kind, "first_differing_ops": found, "seconds": round(time.time() - t0, 1)}), flush=True)
