#!/usr/bin/env python3
"""Run the bf16 engine forward repeatedly on one 3-line batch and count distinct outputs, per engine switch (which kernel is not
bit-reproducible from call to call?)."""
import os, sys, hashlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dtlr_amd import synth, weights
from dtlr_amd.config import DTLRConfig
from dtlr_amd.engine import DTLREngine
dev = torch.device("cuda:0")
cfg = DTLRConfig.latin()
sd = weights.synthetic_state_dict(cfg, seed=0)
x = torch.stack(synth.noise_lines(3, 128, 2048, seed=1000)).to(dev)
mask = torch.zeros((3, 128, 2048), dtype=torch.bool, device=dev)
def digest(t): return hashlib.md5(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:8]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for name, flags in [("default", {}), ("k256 off", {"use_k256": False, "use_k256_small": False}), ("kres off", {"use_kres": False}),
                    ("lds msda off", {"use_lds_msda": False}), ("fused ffn off", {"use_fused_ffn": False}), ("pln k256 off", {"use_pln_k256": False})]:
    if only and name not in only: continue
    eng = DTLREngine(cfg, sd, dev, torch.bfloat16)
    for k, v in flags.items(): setattr(eng, k, v)
    seen = {}
    stages = {}
    for i in range(N):
        out = eng.forward(x, mask, has_padding=False, return_debug=True)
        d = out["_debug"]
        key = digest(out["pred_logits"])
        seen[key] = seen.get(key, 0) + 1
        for st in ("memory", "topk_scores"):
            if st in d: stages.setdefault(st, set()).add(digest(d[st]))
        if "srcs" in d: stages.setdefault("srcs", set()).add("".join(digest(t) for t in d["srcs"]))
    print(f"{name:16s} distinct pred_logits over {N} forwards: {len(seen)}  {sorted(seen.values(), reverse=True)}  stages distinct: { {k: len(v) for k, v in stages.items()} }", flush=True)
