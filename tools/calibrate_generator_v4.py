#!/usr/bin/env python3
"""Calibration of generator v4's detector bank (dtlr_amd/weights.py, GENERATOR v4; CPU only, uses the oracle as the forward).

    python tools/calibrate_generator_v4.py [--lines 8] [--seed 4242]

Builds the v4 weights WITHOUT the detector bank (identical content queries, v2 everywhere else), runs the CPU oracle on a seeded batch of
128x2048 noise lines that is NOT the bench batch, captures the stream entering the last decoder layer's FFN (post-norm1) for every query and
prints, for each name-seeded detector direction u_k, the (1 - V4_TAIL) quantile and the standard deviation of <u_k, x>: the constants frozen in
weights.V4_CALIBRATION.  The numbers are properties of the fp32 network on that batch; they are inputs of the generator, not checked outputs."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dtlr_amd import synth, weights  # noqa: E402
from dtlr_amd.config import DTLRConfig  # noqa: E402
from oracle import dtlr_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=8)
    ap.add_argument("--seed", type=int, default=4242)
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    cfg = DTLRConfig.latin()
    weights._V4_ALLOW_UNCALIBRATED = True
    saved = weights.V4_CALIBRATION.pop((cfg.num_classes, cfg.backbone, 0), None)
    sd = weights.synthetic_state_dict(cfg, 0, version=4)
    if saved is not None:
        weights.V4_CALIBRATION[(cfg.num_classes, cfg.backbone, 0)] = saved
    imgs = synth.noise_lines(args.lines, 128, 2048, seed=args.seed)
    cap = {}
    orig = O.layer_norm

    def hook(sd_, p, x):
        y = orig(sd_, p, x)
        cap[p] = y
        return y
    O.layer_norm = hook
    O.dino_forward(sd, cfg, torch.stack(imgs), mask=torch.zeros(args.lines, 128, 2048, dtype=torch.bool))
    O.layer_norm = orig
    x = cap[f"transformer.decoder.layers.{cfg.dec_layers - 1}.norm1"].reshape(-1, cfg.hidden_dim)
    U, classes = weights.v4_detectors(cfg, 0)
    P = x @ torch.from_numpy(U).t()
    theta = torch.quantile(P, 1 - weights.V4_TAIL, dim=0)
    sigma = P.std(0)
    print(f"# {x.shape[0]} queries ({args.lines} noise lines, seed {args.seed}); V4_TAIL = {weights.V4_TAIL}")
    print(f'V4_CALIBRATION[({cfg.num_classes}, "{cfg.backbone}", 0)] = {{')
    print('    "theta": [' + ", ".join(f"{v:.6f}" for v in theta.tolist()) + "],")
    print('    "sigma": [' + ", ".join(f"{v:.6f}" for v in sigma.tolist()) + "],")
    print("}")
    print("# detector classes:", classes.tolist())


if __name__ == "__main__":
    main()
