#!/usr/bin/env python3
"""Contention-only nondeterminism, second probe (run two copies at once).  Hypothesis under test: a result depends on WHEN a freed
block is re-used.  Per mode, N forwards of the same batch; distinct digests of the logits are counted.  Modes keep chosen tensors
alive until the end of the forward (so their blocks cannot be recycled inside it):
  none        nothing (the failing configuration)
  all         every argument and result of every dtlr_amd.ops call
  msda_args   the arguments of msda_fused only (ow, ref, value view)
  msda_out    the results of msda_fused only
  lin_out     the results of ops.linear only
  not_msda    everything except msda_fused's arguments and results"""
import os, sys, hashlib, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import ops, synth, weights
from dtlr_amd.config import DTLRConfig
from dtlr_amd.engine import DTLREngine
dev = torch.device("cuda:0")
cfg = DTLRConfig.latin()
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[os.environ.get("PROBE_DTYPE", "bf16")]
eng = DTLREngine(cfg, weights.synthetic_state_dict(cfg, seed=0), dev, DT)
if os.environ.get("PROBE_GATHER_ENC") == "1":
    eng.use_lds_msda = False
B = 3
x = torch.stack(synth.noise_lines(B, 128, 2048, seed=1000)).to(dev)
mask = torch.zeros((B, 128, 2048), dtype=torch.bool, device=dev)
keep = []
MODE = "none"
names = [n for n in dir(ops) if callable(getattr(ops, n)) and not n.startswith("_") and n not in ("require_cuda", "msda_encoder_far_fraction", "msda_encoder_fits")
         and getattr(getattr(ops, n), "__module__", "") == "dtlr_amd.ops" and not isinstance(getattr(ops, n), type)]
for n in names:
    fn = getattr(ops, n)
    def wrap(fn=fn, n=n):
        def w(*a, **k):
            r = fn(*a, **k)
            m = MODE
            if m == "all" or (m == "not_msda" and n != "msda_fused"): keep.append((a, k, r))
            elif m == "msda_args" and n == "msda_fused": keep.append((a, k))
            elif m == "msda_out" and n == "msda_fused": keep.append(r)
            elif m == "lin_out" and n == "linear": keep.append(r)
            return r
        return w
    setattr(ops, n, wrap())
def digest(t): return hashlib.md5(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:8]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
tag = sys.argv[2] if len(sys.argv) > 2 else "p"
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["none", "all", "msda_args", "msda_out", "lin_out", "not_msda", "none"]
eng.forward(x, mask, has_padding=False); torch.cuda.synchronize()
for MODE in modes:
    seen = {}
    t0 = time.time()
    for i in range(N):
        keep.clear()
        out = eng.forward(x, mask, has_padding=False)
        key = digest(out["pred_logits"])
        seen[key] = seen.get(key, 0) + 1
    print(f"{tag} mode {MODE:10s}: distinct logits over {N} forwards: {len(seen)} {sorted(seen.values(), reverse=True)} ({time.time() - t0:.1f}s)", flush=True)
