"""Worker of test_two_processes_on_one_gpu_are_bit_reproducible (tests/test_gpu_model.py): N forwards of one seeded batch through the
16-bit engine, back to back, no synchronisation inside a forward; prints how many DISTINCT results it saw and the digest of the first."""
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
    n, dt = int(sys.argv[1]), {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[2]]
    dev = torch.device("cuda:0")
    cfg = DTLRConfig.latin()
    eng = DTLREngine(cfg, weights.synthetic_state_dict(cfg, seed=0), dev, dt)
    x = torch.stack(synth.noise_lines(3, 128, 2048, seed=1000)).to(dev)
    mask = torch.zeros((3, 128, 2048), dtype=torch.bool, device=dev)
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
