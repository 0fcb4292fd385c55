#!/usr/bin/env python3
"""Which operator's output changes from one forward to the next (same input)?  Wraps every dtlr_amd.ops function, hashes its result per
call, and reports the first call whose hash differs from the first forward's.  Run two copies at once: the differences only show under
contention for the GPU."""
import os, sys, hashlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import ops, synth, weights
from dtlr_amd.config import DTLRConfig
from dtlr_amd.engine import DTLREngine
dev = torch.device("cuda:0")
cfg = DTLRConfig.latin()
eng = DTLREngine(cfg, weights.synthetic_state_dict(cfg, seed=0), dev, torch.bfloat16)
x = torch.stack(synth.noise_lines(3, 128, 2048, seed=1000)).to(dev)
mask = torch.zeros((3, 128, 2048), dtype=torch.bool, device=dev)
eng.forward(x, mask, has_padding=False)
log = []
def digest(r):
    ts = r if isinstance(r, (tuple, list)) else (r,)
    h = hashlib.md5()
    for t in ts:
        if torch.is_tensor(t): h.update(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes())
    return h.hexdigest()[:10]
names = [n for n in dir(ops) if callable(getattr(ops, n)) and not n.startswith("_") and n not in ("require_cuda", "msda_encoder_far_fraction", "msda_encoder_fits")
         and getattr(getattr(ops, n), "__module__", "") == "dtlr_amd.ops" and not isinstance(getattr(ops, n), type)]
for n in names:
    fn = getattr(ops, n)
    def wrap(fn=fn, n=n):
        def w(*a, **k):
            r = fn(*a, **k)
            if torch.is_tensor(r): log.append((n, r.detach().clone()))          # device copy, no synchronisation
            elif isinstance(r, (tuple, list)): log.append((n, tuple(t.detach().clone() for t in r if torch.is_tensor(t))))
            return r
        return w
    setattr(ops, n, wrap())
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ref = None
found = {}
for i in range(N):
    log.clear()
    eng.forward(x, mask, has_padding=False)
    torch.cuda.synchronize()
    cur = [(n, digest(r)) for n, r in log]
    if ref is None: ref = cur; print("ops per forward:", len(ref), flush=True); continue
    for j, (a, b) in enumerate(zip(ref, cur)):
        if a != b:
            key = (j, a[0])
            found[key] = found.get(key, 0) + 1
            break
print("first differing op per forward (index, name) -> count:", found, flush=True)
