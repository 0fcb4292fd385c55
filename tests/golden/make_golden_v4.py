"""G9: the REAL reference (imported here, authoring container only) on generator-v4 weights -- the free-running parity set -- for four lines of
bench.py's batch.

    python -m tests.golden.make_golden_v4         # writes tests/golden/g9_v4_free.npz

Generator v4 (dtlr_amd/weights.py) gives every content query the same vector and reads the characters from the image, so the decoded string
does not depend on the rank order of near-tied two-stage scores: the reference's OWN free-running decode is then something another
implementation can be held to without teacher forcing.  Stored: the reference's two-stage scores and selection, per-query top-8 logits, boxes,
and its blank-decoder decision per query in reading order (SetCriterion.loss_CTC's construction, models/dino/dino.py:466-502, eps 0.003).
Inputs are not stored: lines BENCH_ROWS of synth.noise_lines(32, 128, 2048, seed=1000), weights synthetic_state_dict(latin, 0, version=4)."""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from dtlr_amd.config import DTLRConfig                      # noqa: E402
from dtlr_amd.synth import noise_lines                       # noqa: E402
from dtlr_amd.weights import synthetic_state_dict            # noqa: E402
from tests.golden import ref_harness as rh                   # noqa: E402
from tests.golden.make_golden import _decode_with_reference, _np, _run_reference, _top8      # noqa: E402

warnings.filterwarnings("ignore")
BENCH_ROWS = [0, 10, 21, 31]


def main():
    assert rh.reference_available(), "needs /root/reference"
    torch.set_num_threads(16)
    cfg = DTLRConfig.latin()
    sd = synthetic_state_dict(cfg, seed=0, version=4)
    lines = noise_lines(32, 128, 2048, seed=1000)
    imgs = [lines[r] for r in BENCH_ROWS]
    model, post, crit, ref, scores, topk, memory = _run_reference(cfg, sd, imgs)
    v, i = _top8(ref["pred_logits"])
    out = dict(generator_version=np.int64(4), weight_seed=np.int64(0), rows=np.asarray(BENCH_ROWS, dtype=np.int64),
               top8_val=v, top8_idx=i, pred_boxes=_np(ref["pred_boxes"]), topk_idx=_np(topk).astype(np.int16), topk_scores=_np(scores))
    d, _ = _decode_with_reference(ref, post, crit, cfg)
    out["ctc_argmax_eps003"] = d["ctc_argmax_eps003"]
    np.savez_compressed(os.path.join(HERE, "g9_v4_free.npz"), **out)
    seq = torch.from_numpy(out["ctc_argmax_eps003"].astype(np.int64))
    print("g9 written; characters per line:", [(int((s > 0).sum())) for s in seq])


if __name__ == "__main__":
    main()
