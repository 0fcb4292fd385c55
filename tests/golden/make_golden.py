"""Generate the committed golden fixtures by running the REAL reference (authoring container only).

    python -m tests.golden.make_golden            # writes tests/golden/*.npz

Fixtures are data only: seeds/shapes of the synthetic inputs, the reference's outputs (or top-k
subsets of them), the reference's discrete selections, decoded label sequences.  Weights are not
stored: they come from dtlr_amd.weights.synthetic_state_dict (GENERATOR_VERSION recorded).

G1 op level    : MSDeformAttn forward == reference ms_deform_attn_core_pytorch
                 (ops/functions/ms_deform_attn_func.py:41-61), at the ops/test.py shape (seed 3) and
                 at a hot-path-like shape with out-of-range locations.
G2 tiny model  : full DINO.forward + PostProcess + loss_CTC blank construction on a reduced network
                 (32x256 mixed-width pair), every output stored in full.
G3 full model  : Latin_CTC config at 128x2048 (one mixed-width pair), Chinese config at 128x2560;
                 per-query top-8 logits, all boxes, two-stage indices, decoded sequences.
"""
from __future__ import annotations

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

warnings.filterwarnings("ignore")


def _np(t):
    return t.detach().cpu().numpy()


def msda_inputs(N, M, D, Lq, P, shapes, seed, lo=0.0, hi=1.0, value_scale=1.0):
    """Seeded op-level inputs, numpy PCG64 (platform independent)."""
    r = np.random.Generator(np.random.PCG64(seed))
    S = sum(h * w for h, w in shapes)
    L = len(shapes)
    value = (r.random((N, S, M, D), dtype=np.float32) * 2 - 1) * value_scale
    loc = r.uniform(lo, hi, (N, Lq, M, L, P, 2)).astype(np.float32)
    aw = r.random((N, Lq, M, L, P), dtype=np.float32) + 1e-5
    aw = aw / aw.sum((-1, -2), keepdims=True)
    return torch.from_numpy(value), torch.as_tensor(shapes, dtype=torch.long), torch.from_numpy(loc), torch.from_numpy(aw.astype(np.float32))


def gen_g1():
    core = rh.reference_core_pytorch()
    out = {}
    # (a) the reference's own unit-test shape (ops/test.py:21-28)
    v, s, loc, aw = msda_inputs(1, 2, 2, 2, 2, [(6, 4), (3, 2)], seed=3, value_scale=0.01)
    out["a_value"], out["a_shapes"], out["a_loc"], out["a_aw"] = _np(v), _np(s), _np(loc), _np(aw)
    out["a_out_f32"] = _np(core(v, s, loc, aw))
    out["a_out_f64"] = _np(core(v.double(), s, loc.double(), aw.double()))
    # (b) hot-path-like: 4 levels of a 32x512 line, 8 heads x 32 ch, locations in [-0.5,1.5]
    shapes = [(4, 64), (2, 32), (1, 16), (1, 8)]
    v, s, loc, aw = msda_inputs(2, 8, 32, 77, 4, shapes, seed=11, lo=-0.5, hi=1.5)
    out["b_seed"], out["b_shapes"] = np.int64(11), _np(s)
    out["b_out_f32"] = _np(core(v, s, loc, aw))
    # (c) the real encoder shape (S=Lq=5440) for one line: store a strided subsample + checksum
    shapes = [(16, 256), (8, 128), (4, 64), (2, 32)]
    v, s, loc, aw = msda_inputs(1, 8, 32, 5440, 4, shapes, seed=12, lo=-0.25, hi=1.25)
    o = core(v, s, loc, aw)
    out["c_seed"], out["c_shapes"] = np.int64(12), _np(s)
    out["c_out_rows"] = _np(o[0, ::97])
    out["c_out_sum"] = np.float64(o.double().sum().item())
    out["c_out_abs_sum"] = np.float64(o.double().abs().sum().item())
    np.savez_compressed(os.path.join(HERE, "g1_msda.npz"), **out)
    print("g1 written")


def _run_reference(cfg, sd, imgs):
    model, post, crit = rh.build_reference_model(cfg, sd)
    cap = {}
    def grab_cls(m, i, o):            # hooks must return None (a value would replace the output)
        if "cls" not in cap:          # 1st call = all S tokens (deformable_transformer.py:341); the
            cap["cls"] = o.detach()   # 2nd call is the interm head on the selected rows (dino.py:371)

    def grab_mem(m, i, o):
        cap["memory"] = o[0].detach()

    h1 = model.transformer.enc_out_class_embed.register_forward_hook(grab_cls)
    h2 = model.transformer.encoder.register_forward_hook(grab_mem)
    with torch.no_grad():
        ref = model(imgs)
    h1.remove(), h2.remove()
    scores = cap["cls"].max(-1)[0]
    topk = torch.topk(scores, cfg.num_queries, dim=1)[1]
    return model, post, crit, ref, scores, topk, cap["memory"]


def _decode_with_reference(ref, post, crit, cfg):
    """Run the reference's own PostProcess / loss_CTC blank construction on its outputs."""
    B = ref["pred_logits"].shape[0]
    ts = torch.ones(B, 2)
    d = {}
    post["bbox"].num_select, post["bbox"].nms_iou_threshold = cfg.num_select, -1
    r = post["bbox"](ref, ts)
    d["pp_scores"] = _np(torch.stack([x["scores"] for x in r]))
    d["pp_labels"] = _np(torch.stack([x["labels"] for x in r]))
    d["pp_boxes"] = _np(torch.stack([x["boxes"] for x in r]))
    tg = [{"labels": torch.tensor([1, 2])} for _ in range(B)]
    _, newp, _ = crit.loss_CTC(ref, tg, None, None, return_preds=True)          # eps = 0.003
    d["ctc_argmax_eps003"] = _np(newp.argmax(-1)).astype(np.int32)
    return d, newp


def gen_g2():
    cfg = DTLRConfig.tiny()
    sd = synthetic_state_dict(cfg, seed=0)
    imgs = stroke_lines(1, 32, 256, seed=5) + noise_lines(1, 32, 192, seed=6)
    model, post, crit, ref, scores, topk, memory = _run_reference(cfg, sd, imgs)
    out = dict(generator_version=np.int64(GENERATOR_VERSION), weight_seed=np.int64(0),
               pred_logits=_np(ref["pred_logits"]), pred_boxes=_np(ref["pred_boxes"]),
               interm_logits=_np(ref["interm_outputs"]["pred_logits"]), interm_boxes=_np(ref["interm_outputs"]["pred_boxes"]),
               init_box_proposal=_np(ref["interm_outputs_for_matching_pre"]["pred_boxes"]),
               aux_logits=_np(torch.stack([a["pred_logits"] for a in ref["aux_outputs"]])),
               aux_boxes=_np(torch.stack([a["pred_boxes"] for a in ref["aux_outputs"]])),
               topk_idx=_np(topk).astype(np.int32), topk_scores=_np(scores), memory=_np(memory))
    d, newp = _decode_with_reference(ref, post, crit, cfg)
    out.update(d)
    out["ctc_probs_eps003"] = _np(newp)
    # NMS PostProcess as the NMS decoder drives it (evaluation.py:97-100), per sample
    post["bbox"].num_select, post["bbox"].nms_iou_threshold = cfg.num_queries, 0.5
    for b in range(2):
        one = {"pred_logits": ref["pred_logits"][b:b + 1], "pred_boxes": ref["pred_boxes"][b:b + 1]}
        r = post["bbox"](one, torch.tensor([[1.0, 1.0]]))[0]
        out[f"nms_labels_{b}"], out[f"nms_scores_{b}"] = _np(r["labels"]).astype(np.int32), _np(r["scores"])
    np.savez_compressed(os.path.join(HERE, "g2_tiny_model.npz"), **out)
    print("g2 written")


def _top8(logits):
    v, i = torch.topk(logits, 8, dim=-1)
    return _np(v), _np(i).astype(np.int16)


def gen_g3():
    for tag, cfg, widths, h in (("latin", DTLRConfig.latin(), (2048, 1536), 128),
                                ("chinese", DTLRConfig.chinese(), (2560, 1792), 128)):
        sd = synthetic_state_dict(cfg, seed=0)
        imgs = stroke_lines(1, h, widths[0], seed=21) + noise_lines(1, h, widths[1], seed=22)
        model, post, crit, ref, scores, topk, memory = _run_reference(cfg, sd, imgs)
        v, i = _top8(ref["pred_logits"])
        out = dict(generator_version=np.int64(GENERATOR_VERSION), weight_seed=np.int64(0),
                   height=np.int64(h), widths=np.asarray(widths, dtype=np.int64),
                   top8_val=v, top8_idx=i, logits_rowsum=_np(ref["pred_logits"].double().sum(-1)),
                   pred_boxes=_np(ref["pred_boxes"]), topk_idx=_np(topk).astype(np.int16),
                   topk_scores=_np(scores), memory_rows=_np(memory[:, ::67]))
        d, _ = _decode_with_reference(ref, post, crit, cfg)
        out.update(d)
        # the NMS decoder's PostProcess call (evaluation.py:97-100: num_select = 900, IoU 0.5, target size (1,1)), per sample
        post["bbox"].num_select, post["bbox"].nms_iou_threshold = cfg.num_queries, 0.5
        for b in range(2):
            one = {"pred_logits": ref["pred_logits"][b:b + 1], "pred_boxes": ref["pred_boxes"][b:b + 1]}
            r = post["bbox"](one, torch.tensor([[1.0, 1.0]]))[0]
            out[f"nms_labels_{b}"], out[f"nms_scores_{b}"] = _np(r["labels"]).astype(np.int32), _np(r["scores"])
            out[f"nms_boxes_{b}"] = _np(r["boxes"])
        # per-query decision data for margin-aware gates: the two largest logits of every query are in top8_val[..., :2]
        np.savez_compressed(os.path.join(HERE, f"g3_{tag}.npz"), **out)
        print("g3", tag, "written")
        del model, sd


if __name__ == "__main__":
    assert rh.reference_available(), "needs /root/reference"
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["g1", "g2", "g3"]
    if "g1" in which:
        gen_g1()
    if "g2" in which:
        gen_g2()
    if "g3" in which:
        gen_g3()
