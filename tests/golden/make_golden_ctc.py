#!/usr/bin/env python3
"""Golden vectors for the evaluation-time CTC loss value: tests/golden/g5_ctc.npz.

Runs the REAL reference criterion (`SetCriterion.loss_CTC`, models/dino/dino.py:457-551, through tests/golden/ref_harness.py)
on seeded synthetic head outputs and label sequences; stores the seeds, the label sequences and the loss values.
Authoring container only:  python tests/golden/make_golden_ctc.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from dtlr_amd.config import DTLRConfig                      # noqa: E402
from dtlr_amd.weights import synthetic_state_dict           # noqa: E402
from tests.golden import ref_harness as rh                  # noqa: E402
from tests.util import ctc_case                             # noqa: E402

# (seed, B, nq, C, bias, max target length)
CASES = [(1, 3, 30, 23, -3.0, 12), (2, 2, 30, 23, -1.0, 20), (3, 2, 900, 166, -6.0, 80), (4, 4, 64, 11, -2.0, 40),
         (5, 2, 30, 23, -8.0, 29)]


def main():
    cfg = DTLRConfig.tiny()
    _, _, crit = rh.build_reference_model(cfg, synthetic_state_dict(cfg, 0))
    out = {"cases": np.array(CASES, dtype=np.float64)}
    for k, (seed, B, nq, C, bias, lmax) in enumerate(CASES):
        outputs, labels = ctc_case(seed, B, nq, C, bias, lmax)
        targets = [{"labels": torch.tensor(l, dtype=torch.int64)} for l in labels]
        with torch.no_grad():
            loss = crit.loss_CTC(outputs, targets, None, None)["loss_CTC"]
        out[f"loss_{k}"] = np.float64(loss.item())
        print(k, (seed, B, nq, C, bias, lmax), float(loss))
    np.savez_compressed(os.path.join(HERE, "g5_ctc.npz"), **out)


if __name__ == "__main__":
    main()
