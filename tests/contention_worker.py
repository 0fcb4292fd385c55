"""Worker of test_two_processes_on_one_gpu_are_bit_reproducible (tests/test_gpu_model.py): N forwards of one seeded batch through the
chosen engine, back to back, no synchronisation inside a forward; prints how many DISTINCT results it saw and the digest of the first."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dtlr_amd import synth, weights  # noqa: E402
from dtlr_amd.config import DTLRConfig  # noqa: E402
from dtlr_amd.engine import DTLREngine  # noqa: E402


def main():
    """argv: <forwards> <engine> [config]   engine: bf16 | f16 | f32 | f32s | bf16-gather (the encoder forced onto the gather kernel);
    config: latin (default) | chinese (7356-class head, B = 2) | swin (Latin head on swin_T_224_1k)"""
    n, kind = int(sys.argv[1]), sys.argv[2]
    conf = sys.argv[3] if len(sys.argv) > 3 else "latin"
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "bf16-gather": torch.bfloat16}.get(kind, torch.float32)
    dev = torch.device("cuda:0")
    cfg = DTLRConfig.chinese() if conf == "chinese" else DTLRConfig.latin()
    if conf == "swin":
        import dataclasses
        cfg = dataclasses.replace(cfg, backbone="swin_T_224_1k")
    eng = DTLREngine(cfg, weights.synthetic_state_dict(cfg, seed=0), dev, dt, split=kind == "f32s")
    if kind == "bf16-gather":
        eng.use_lds_msda = False
    B, W = (2, 2560) if conf == "chinese" else (3, 2048)
    x = torch.stack(synth.noise_lines(B, 128, W, seed=1000)).to(dev)
    mask = torch.zeros((B, 128, W), dtype=torch.bool, device=dev)
    seen = {}
    for _ in range(n):
        out = eng.forward(x, mask, has_padding=False)
        h = hashlib.md5()
        for k in ("pred_logits", "pred_boxes"):
            h.update(out[k].float().cpu().numpy().tobytes())
        seen[h.hexdigest()] = seen.get(h.hexdigest(), 0) + 1
    print(json.dumps({"forwards": n, "distinct": len(seen), "digest": max(seen, key=seen.get)}), flush=True)


if __name__ == "__main__":
    main()
