"""Golden fixtures for the Swin backbones, from the REAL reference (authoring container only).

    python -m tests.golden.make_golden_swin          # writes tests/golden/g7_swin.npz

G7a backbone level: the reference's SwinTransformer class (models/dino/swin_transformer.py:435) instantiated with a test-sized
    network (embed 32, depths 2/2/2/2, heads 1/2/4/8, window 4) on an odd-sized pair (37 x 90: patch padding, window padding,
    odd patch-merging sizes, shifted windows), name-seeded weights of dtlr_amd.weights (cfg.backbone = "swin_custom").
G7b full model: reference build_dino with backbone = 'swin_T_224_1k' (backbone.py:172-205) and a 2+2-layer transformer on a
    mixed-width pair: logits (top-8), boxes, two-stage selection, memory rows.
"""
from __future__ import annotations

import dataclasses
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from dtlr_amd.config import DTLRConfig                      # noqa: E402
from dtlr_amd.synth import noise_lines, stroke_lines         # noqa: E402
from dtlr_amd.weights import GENERATOR_VERSION, synthetic_state_dict   # noqa: E402
from tests.golden import ref_harness as rh                   # noqa: E402
from tests.golden.make_golden import _run_reference, _top8, _np   # noqa: E402

warnings.filterwarnings("ignore")


def custom_cfg():
    return dataclasses.replace(DTLRConfig.tiny(), backbone="swin_custom", swin_embed_dim=32, swin_depths=(2, 2, 2, 2),
                               swin_num_heads=(1, 2, 4, 8), swin_window=4)


def swin_t_cfg():
    return dataclasses.replace(DTLRConfig.tiny(), backbone="swin_T_224_1k")


def main():
    rh._install_stubs()
    from models.dino.swin_transformer import SwinTransformer
    out = {"generator_version": np.int64(GENERATOR_VERSION)}
    # ---- G7a
    cfg = custom_cfg()
    sd = synthetic_state_dict(cfg, seed=0)
    m = SwinTransformer(pretrain_img_size=224, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=4,
                        out_indices=(1, 2, 3), drop_path_rate=0.0)
    m.eval()
    missing, unexpected = m.load_state_dict({k[len("backbone.0."):]: v for k, v in sd.items() if k.startswith("backbone.0.")}, strict=True)
    x = torch.stack(noise_lines(2, 37, 90, seed=71))
    with torch.no_grad():
        feats = m.forward_raw(x)
    for i, f in enumerate(feats):
        out[f"a_feat{i}"] = _np(f)
    # ---- G7b
    cfg = swin_t_cfg()
    sd = synthetic_state_dict(cfg, seed=0)
    imgs = stroke_lines(1, 64, 256, seed=5) + noise_lines(1, 48, 200, seed=6)
    model, post, crit, ref, scores, topk, memory = _run_reference(cfg, sd, imgs)
    v, i = _top8(ref["pred_logits"])
    out.update(b_top8_val=v, b_top8_idx=i, b_pred_boxes=_np(ref["pred_boxes"]), b_topk_idx=_np(topk).astype(np.int16),
               b_topk_scores=_np(scores), b_memory=_np(memory[:, ::7]), b_pred_logits=_np(ref["pred_logits"]))
    np.savez_compressed(os.path.join(HERE, "g7_swin.npz"), **out)
    print("g7 written")


if __name__ == "__main__":
    assert rh.reference_available(), "needs /root/reference"
    torch.set_num_threads(8)
    main()
