"""Pin the oracle against the committed golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from dtlr_amd.config import DTLRConfig
from dtlr_amd.synth import noise_lines, stroke_lines
from dtlr_amd.weights import GENERATOR_VERSION, synthetic_state_dict
from oracle import dtlr_oracle as O
from tests.util import c_oracle_msda, msda_inputs


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_g1_msda_reference_unit_test_shape(golden_dir, oracle_clib):
    """ops/test.py shape (N,M,D=1,2,2; Lq,L,P=2,2,2; shapes (6,4),(3,2)), tolerances of ops/test.py:40,56."""
    g = np.load(os.path.join(golden_dir, "g1_msda.npz"))
    v, s, loc, aw = _t(g["a_value"]), _t(g["a_shapes"]), _t(g["a_loc"]), _t(g["a_aw"])
    lsi = torch.cat((s.new_zeros((1,)), s.prod(1).cumsum(0)[:-1]))
    o32 = O.ms_deform_attn_core(v, s, loc, aw)
    assert torch.allclose(o32, _t(g["a_out_f32"]), rtol=1e-2, atol=1e-3)
    assert (o32 - _t(g["a_out_f32"])).abs().max() < 1e-7
    o64 = O.ms_deform_attn_core(v.double(), s, loc.double(), aw.double())
    assert torch.allclose(o64, _t(g["a_out_f64"]))
    c32 = c_oracle_msda(oracle_clib, v, s, lsi, loc, aw)
    assert (c32 - _t(g["a_out_f32"])).abs().max() < 1e-7
    c64 = c_oracle_msda(oracle_clib, v.double(), s, lsi, loc.double(), aw.double())
    assert torch.allclose(c64, _t(g["a_out_f64"]))


def test_g1_msda_out_of_range_locations(golden_dir, oracle_clib):
    g = np.load(os.path.join(golden_dir, "g1_msda.npz"))
    shapes = [tuple(x) for x in g["b_shapes"].tolist()]
    v, s, lsi, loc, aw = msda_inputs(2, 8, 32, 77, 4, shapes, seed=int(g["b_seed"]), lo=-0.5, hi=1.5)
    ref = _t(g["b_out_f32"])
    assert (O.ms_deform_attn_core(v, s, loc, aw) - ref).abs().max() < 2e-6
    # the grid_sample formulation (the reference's own CPU core, used by bench.py's cpu_baseline leg) is the same function
    assert (O.ms_deform_attn_core_grid_sample(v, s, loc, aw) - ref).abs().max() < 2e-6
    assert (c_oracle_msda(oracle_clib, v, s, lsi, loc, aw) - ref).abs().max() < 2e-6


def test_g1_msda_encoder_shape(golden_dir, oracle_clib):
    g = np.load(os.path.join(golden_dir, "g1_msda.npz"))
    shapes = [tuple(x) for x in g["c_shapes"].tolist()]
    v, s, lsi, loc, aw = msda_inputs(1, 8, 32, 5440, 4, shapes, seed=int(g["c_seed"]), lo=-0.25, hi=1.25)
    o = c_oracle_msda(oracle_clib, v, s, lsi, loc, aw)
    assert (o[0, ::97] - _t(g["c_out_rows"])).abs().max() < 2e-6
    assert abs(o.double().sum().item() - float(g["c_out_sum"])) < 1e-2
    assert abs(o.double().abs().sum().item() - float(g["c_out_abs_sum"])) / float(g["c_out_abs_sum"]) < 1e-6
    o2 = O.ms_deform_attn_core(v, s, loc, aw)
    assert (o2 - o).abs().max() < 2e-6


def test_g2_tiny_model_full_outputs(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_tiny_model.npz"))
    assert int(g["generator_version"]) == GENERATOR_VERSION
    cfg = DTLRConfig.tiny()
    sd = synthetic_state_dict(cfg, seed=int(g["weight_seed"]))
    imgs = stroke_lines(1, 32, 256, seed=5) + noise_lines(1, 32, 192, seed=6)
    out = O.dino_forward(sd, cfg, imgs, forced_topk=_t(g["topk_idx"]).long(), return_debug=True)
    d = out["_debug"]
    assert (d["memory"] - _t(g["memory"])).abs().max() < 1e-4
    assert (d["topk_scores"] - _t(g["topk_scores"])).abs().max() < 1e-4
    # free-running selection agrees here (no near ties at this size)
    free = torch.topk(d["topk_scores"], cfg.num_queries, dim=1)[1]
    assert torch.equal(free, _t(g["topk_idx"]).long())
    assert (out["pred_logits"] - _t(g["pred_logits"])).abs().max() < 1e-3        # the north_star tolerance
    assert (out["pred_logits"] - _t(g["pred_logits"])).abs().max() < 1e-4
    assert (out["pred_boxes"] - _t(g["pred_boxes"])).abs().max() < 1e-5
    assert (out["interm_outputs"]["pred_logits"] - _t(g["interm_logits"])).abs().max() < 1e-4
    assert (out["interm_outputs"]["pred_boxes"] - _t(g["interm_boxes"])).abs().max() < 1e-5
    assert (out["interm_outputs_for_matching_pre"]["pred_boxes"] - _t(g["init_box_proposal"])).abs().max() < 1e-5
    aux_l = torch.stack([a["pred_logits"] for a in out["aux_outputs"]])
    aux_b = torch.stack([a["pred_boxes"] for a in out["aux_outputs"]])
    assert (aux_l - _t(g["aux_logits"])).abs().max() < 1e-4
    assert (aux_b - _t(g["aux_boxes"])).abs().max() < 1e-5


def test_g2_decoders_on_reference_outputs(golden_dir):
    """PostProcess / blank construction / NMS applied to the REFERENCE's logits must reproduce the
    reference's own results exactly (same inputs -> discrete outputs identical)."""
    g = np.load(os.path.join(golden_dir, "g2_tiny_model.npz"))
    cfg = DTLRConfig.tiny()
    ref = {"pred_logits": _t(g["pred_logits"]), "pred_boxes": _t(g["pred_boxes"])}
    pp = O.post_process(ref, torch.ones(2, 2), cfg.num_select)
    assert torch.equal(torch.stack([p["labels"] for p in pp]), _t(g["pp_labels"]))
    assert (torch.stack([p["scores"] for p in pp]) - _t(g["pp_scores"])).abs().max() < 1e-7
    assert (torch.stack([p["boxes"] for p in pp]) - _t(g["pp_boxes"])).abs().max() < 1e-6
    probs = O.blank_probabilities(ref, 0.003)
    assert (probs - _t(g["ctc_probs_eps003"])).abs().max() < 1e-6
    assert torch.equal(probs.argmax(-1).int(), _t(g["ctc_argmax_eps003"]))
    for b in range(2):
        one = {"pred_logits": ref["pred_logits"][b:b + 1], "pred_boxes": ref["pred_boxes"][b:b + 1]}
        r = O.post_process(one, torch.tensor([[1.0, 1.0]]), cfg.num_queries, 0.5)[0]
        assert torch.equal(r["labels"].int(), _t(g[f"nms_labels_{b}"]))
        assert (r["scores"] - _t(g[f"nms_scores_{b}"])).abs().max() < 1e-7


@pytest.mark.parametrize("tag", ["latin", "chinese"])
def test_g3_full_model(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"g3_{tag}.npz"))
    cfg = DTLRConfig.latin() if tag == "latin" else DTLRConfig.chinese()
    sd = synthetic_state_dict(cfg, seed=int(g["weight_seed"]))
    h, widths = int(g["height"]), [int(w) for w in g["widths"]]
    imgs = stroke_lines(1, h, widths[0], seed=21) + noise_lines(1, h, widths[1], seed=22)
    topk = _t(g["topk_idx"].astype(np.int64))
    out = O.dino_forward(sd, cfg, imgs, forced_topk=topk, return_debug=True)
    d = out["_debug"]
    assert (d["topk_scores"] - _t(g["topk_scores"])).abs().max() < 1e-4
    assert (d["memory"][:, ::67] - _t(g["memory_rows"])).abs().max() < 2e-4
    idx = _t(g["top8_idx"].astype(np.int64))
    assert (torch.gather(out["pred_logits"], 2, idx) - _t(g["top8_val"])).abs().max() < 1e-3
    assert (out["pred_logits"].double().sum(-1) - _t(g["logits_rowsum"])).abs().max() < 1e-3 * cfg.num_classes
    assert (out["pred_boxes"] - _t(g["pred_boxes"])).abs().max() < 1e-5


# ---- eval-time preprocessing: the oracle's Pillow-resample restatement -------------------
def test_g4_preproc_resize_matches_pillow_vectors(golden_dir):
    """oracle.pil_resize_bilinear_u8 == the vectors Pillow produced (tests/golden/make_golden_preproc.py), bit for bit;
    the size rule == the reference's (the generating script asserted it against datasets/transforms.py:81-99)."""
    import hashlib
    from tests.util import preproc_image
    g = np.load(os.path.join(golden_dir, "g4_preproc.npz"))
    for k, (seed, h, w, size, max_size, oh, ow) in enumerate(g["small_cases"].tolist()):
        ms = None if max_size < 0 else max_size
        assert O.get_size_with_aspect_ratio((w, h), size, ms) == (oh, ow)
        got = O.pil_resize_bilinear_u8(preproc_image(h, w, seed), oh, ow)
        assert np.array_equal(got, g[f"s{k}_out"]), (seed, h, w, size, ms)
    for (seed, h, w, size, max_size, oh, ow), digest in zip(g["big_cases"].tolist(), g["big_sha256"].tolist()):
        assert O.get_size_with_aspect_ratio((w, h), size, max_size) == (oh, ow)
        got = O.pil_resize_bilinear_u8(preproc_image(h, w, seed), oh, ow)
        assert hashlib.sha256(got.tobytes()).hexdigest() == digest, (seed, h, w)


def test_preproc_oracle_against_installed_pillow():
    """Same check against whatever Pillow is installed where the tests run (skipped without it): random sizes, both
    scaling directions per axis, identity axes."""
    PILImage = pytest.importorskip("PIL.Image")
    g = np.random.Generator(np.random.PCG64(5))
    for _ in range(12):
        h, w = int(g.integers(3, 90)), int(g.integers(3, 400))
        oh, ow = int(g.integers(2, 120)), int(g.integers(2, 500))
        if g.random() < 0.25:
            oh = h
        if g.random() < 0.25:
            ow = w
        img = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(PILImage.fromarray(img, "RGB").resize((ow, oh), PILImage.BILINEAR))
        assert np.array_equal(O.pil_resize_bilinear_u8(img, oh, ow), ref), (h, w, oh, ow)


def test_preproc_pipeline_shapes_and_mask():
    """preprocess_lines == per-item transform + collate (util/misc.py:375-397): canvas = max sizes, zero padding,
    mask True exactly on the padding; values = ((u8/255) - mean) / std."""
    from tests.util import preproc_image
    imgs = [preproc_image(20, 200, 1), preproc_image(31, 90, 2), preproc_image(12, 12, 3)]
    x, m = O.preprocess_lines(imgs, size=16, max_size=64)
    sizes = [O.get_size_with_aspect_ratio((im.shape[1], im.shape[0]), 16, 64) for im in imgs]
    assert tuple(x.shape) == (3, 3, max(s[0] for s in sizes), max(s[1] for s in sizes)) and m.dtype == torch.bool
    for b, (oh, ow) in enumerate(sizes):
        assert not m[b, :oh, :ow].any() and m[b, oh:].all() and m[b, :, ow:].all()
        assert (x[b, :, oh:] == 0).all() and (x[b, :, :, ow:] == 0).all()
        r = torch.from_numpy(O.pil_resize_bilinear_u8(imgs[b], oh, ow)).permute(2, 0, 1).float()
        want = (r / 255 - torch.tensor(O.IMAGENET_MEAN).view(3, 1, 1)) / torch.tensor(O.IMAGENET_STD).view(3, 1, 1)
        assert torch.equal(x[b, :, :oh, :ow], want)


# ---- evaluation-time CTC loss value -----------------------------------------------------------
def test_g5_ctc_loss_matches_reference_criterion(golden_dir):
    """oracle.loss_ctc == SetCriterion.loss_CTC of the real reference (tests/golden/make_golden_ctc.py): same arithmetic
    (torch's CTCLoss on CPU), so the tolerance only absorbs summation-order noise."""
    from tests.util import ctc_case
    g = np.load(os.path.join(golden_dir, "g5_ctc.npz"))
    for k, (seed, B, nq, C, bias, lmax) in enumerate(g["cases"].tolist()):
        outputs, labels = ctc_case(int(seed), int(B), int(nq), int(C), bias, int(lmax))
        got = O.loss_ctc(outputs, labels).item()
        assert abs(got - float(g[f"loss_{k}"])) <= 1e-5 * max(1.0, abs(float(g[f"loss_{k}"]))), (k, got, float(g[f"loss_{k}"]))


# ---- Swin backbones ---------------------------------------------------------------------------
def test_g7_swin_backbone_and_full_model(golden_dir):
    """oracle.swin_body == the reference's SwinTransformer class (test-sized network, odd sizes: patch padding, window padding,
    shifted windows, odd patch merging) and oracle.dino_forward with backbone = swin_T_224_1k == the reference's build_dino
    (tests/golden/make_golden_swin.py)."""
    from tests.golden.make_golden_swin import custom_cfg, swin_t_cfg
    g = np.load(os.path.join(golden_dir, "g7_swin.npz"))
    cfg = custom_cfg()
    sd = synthetic_state_dict(cfg, seed=0)
    x = torch.stack(noise_lines(2, 37, 90, seed=71))
    feats = O.swin_body(x, sd, cfg.swin_params())
    for i, f in enumerate(feats):
        assert (f - _t(g[f"a_feat{i}"])).abs().max() < 2e-5
    cfg = swin_t_cfg()
    sd = synthetic_state_dict(cfg, seed=0)
    imgs = stroke_lines(1, 64, 256, seed=5) + noise_lines(1, 48, 200, seed=6)
    out = O.dino_forward(sd, cfg, imgs, forced_topk=_t(g["b_topk_idx"].astype(np.int64)), return_debug=True)
    assert (out["_debug"]["topk_scores"] - _t(g["b_topk_scores"])).abs().max() < 1e-4
    assert (out["_debug"]["memory"][:, ::7] - _t(g["b_memory"])).abs().max() < 2e-4
    assert (out["pred_logits"] - _t(g["b_pred_logits"])).abs().max() < 1e-3
    assert (out["pred_boxes"] - _t(g["b_pred_boxes"])).abs().max() < 1e-5


def test_tie_aware_compare_counts_what_it_says():
    """oracle/compare.py::tie_aware_compare on constructed decodes: identical inputs give zeros; a label flip is explained exactly when the
    reference margin is below twice the stated logit error; two printing queries that trade places are explained exactly when their
    reference cx differ by less than twice the stated cx error; the edit distance is over the strings of ALL queries."""
    from oracle.compare import query_decisions, tie_aware_compare
    C, nq = 12, 6

    def make(labels, strength, cx):
        """one class logit per query (the others at -12): label l prints when its logit is > 0 (p > 1 - p = the blank channel), the
        decision margin is ~ |logit|; label -1 = every logit at -12 (blank with a large margin)"""
        lg = torch.full((1, nq, C), -12.0)
        bx = torch.zeros((1, nq, 4))
        for i, (l, m, x) in enumerate(zip(labels, strength, cx)):
            if l >= 0:
                lg[0, i, l] = m
            bx[0, i] = torch.tensor([x, 0.5, 0.01, 0.5])
        return lg, bx

    labels = [3, 4, -1, 5, -1, 6]
    st = [2.0, 0.001, 0.0, 2.0, 0.0, 2.0]
    cx = [0.10, 0.30, 0.50, 0.3001, 0.70, 0.90]
    rl, rb = make(labels, st, cx)
    lab, mar = query_decisions(rl, rb, 0.03)
    assert lab[0].tolist() == labels and 5e-4 < float(mar[0, 1]) < 2e-3 and float(mar[0, 0]) > 1.0
    same = tie_aware_compare(rl, rb, rl.clone(), rb.clone(), 0.03, 1e-3, 1e-4)
    assert same["edit_distance"] == 0 and same["label_flips"] == 0 and same["order_swaps"] == 0 and same["unexplained"] == 0
    assert same["chars_ref"] == 4 and abs(same["min_gap_px_2048"] - 0.0001 * 2048) < 1e-2
    # (1) query 1 (margin ~0.001) drops to blank: explained at logit_err 1e-3 (2e-3 > margin), unexplained at 1e-4
    st1 = list(st)
    st1[1] = -0.001
    gl, gb = make(labels, st1, cx)
    r = tie_aware_compare(rl, rb, gl, gb, 0.03, 1e-3, 1e-4)
    assert r["label_flips"] == 1 and r["label_flips_unexplained"] == 0 and r["edit_distance"] == 1 and r["unexplained"] == 0
    r = tie_aware_compare(rl, rb, gl, gb, 0.03, 1e-4, 1e-4)
    assert r["label_flips_unexplained"] == 1 and r["unexplained"] == 1
    # (2) queries 1 and 3 (cx 0.3000 / 0.3001) trade places: explained at cx_err 1e-4 (gap 1e-4 < 2e-4), unexplained at 2e-5
    cx2 = list(cx)
    cx2[1], cx2[3] = 0.30012, 0.30008
    gl, gb = make(labels, st, cx2)
    r = tie_aware_compare(rl, rb, gl, gb, 0.03, 1e-3, 1e-4)
    assert r["order_swaps"] == 1 and r["order_swaps_unexplained"] == 0 and r["label_flips"] == 0 and r["edit_distance"] == 2
    r = tie_aware_compare(rl, rb, gl, gb, 0.03, 1e-3, 2e-5)
    assert r["order_swaps_unexplained"] == 1 and r["unexplained"] == 1
    # (3) a flip of a confident query is never explained
    gl, gb = make([7, 4, -1, 5, -1, 6], st, cx)
    r = tie_aware_compare(rl, rb, gl, gb, 0.03, 1e-2, 1e-4)
    assert r["label_flips"] == 1 and r["label_flips_unexplained"] == 1


def test_oracle_resume_and_parity_legs():
    """oracle/parity.py on a constructed 'engine' (the oracle's own outputs, perturbed): dino_forward(resume=...) reproduces the full
    forward bit for bit (also row-sliced, also teacher-forced); an unperturbed engine is identical on both legs; a logit error beyond
    the fp32 budget fails the gate; a forced rank swap shows up in the free-running leg (rank_slots_changed) while the teacher-forced
    leg, which follows the engine's selection, stays exact."""
    from oracle import dtlr_oracle as O
    from oracle.parity import OracleBatch
    cfg = DTLRConfig.tiny()
    sd = synthetic_state_dict(cfg, 0)
    imgs = stroke_lines(3, 32, 256, seed=9)
    x, m = torch.stack(imgs), torch.zeros(3, 32, 256, dtype=torch.bool)
    ob = OracleBatch(cfg, sd, x, m, threads=4)
    f, d = ob.free, ob.debug
    again = O.dino_forward(sd, cfg, None, resume=d)
    assert torch.equal(again["pred_logits"], f["pred_logits"]) and torch.equal(again["pred_boxes"], f["pred_boxes"])
    sub = O.dino_forward(sd, cfg, None, forced_topk=d["topk_idx"][[0, 2]], resume=O.resume_rows(d, [0, 2]))
    assert torch.equal(sub["pred_logits"], f["pred_logits"][[0, 2]])
    same = ob.compare("f32s", f["pred_logits"], f["pred_boxes"], d["topk_idx"], d["topk_scores"])
    assert same["parity_gate"] and same["free_running"]["strings_identical_free_running"] == "3/3"
    assert same["teacher_forced"]["logit_err_max"] == 0.0 and same["free_running"]["rank_slots_changed"] == 0
    off = ob.compare("f32s", f["pred_logits"] + 2e-3, f["pred_boxes"], d["topk_idx"], d["topk_scores"])
    assert not off["teacher_forced"]["within_budget"] and not off["parity_gate"]
    assert ob.compare("bf16", f["pred_logits"] + 2e-3, f["pred_boxes"], d["topk_idx"], d["topk_scores"])["parity_gate"]      # inside the bf16 budget
    idx = d["topk_idx"].clone()
    idx[:, [0, 1]] = idx[:, [1, 0]]                                # ranks 0 and 1 trade places: the two queries trade tgt_embed rows
    sw = O.dino_forward(sd, cfg, None, forced_topk=idx, resume=d)
    r = ob.compare("f32", sw["pred_logits"], sw["pred_boxes"], idx, d["topk_scores"])
    assert r["free_running"]["rank_slots_changed"] == 6 and r["free_running"]["lines_with_identical_selection"] == 0
    assert r["teacher_forced"]["logit_err_max"] == 0.0 and r["teacher_forced"]["strings_identical_same_selection"] == "3/3"


def test_generator_v4_is_rank_invariant_and_sparse():
    """Generator v4 (dtlr_amd/weights.py; the free-running parity set) through the oracle, one 128x2048 line: it keeps every v2 tensor outside
    tgt_embed and the first V4_DETECTORS units of the last decoder FFN; its content queries are one vector; the line carries text (a few
    dozen characters out of 900 queries) with decisive margins; and the decode is invariant to what any fp32 implementation does to the
    selection: a random PERMUTATION of the selected tokens' ranks and score noise of 1e-4 (hundreds of rank swaps) leave the string unchanged."""
    from dtlr_amd import weights as Wt
    from oracle import dtlr_oracle as O
    from oracle.compare import query_decisions
    from oracle.parity import OracleBatch
    cfg = DTLRConfig.latin()
    sd2, sd4 = synthetic_state_dict(cfg, 0), synthetic_state_dict(cfg, 0, version=4)
    t = "transformer."
    p5 = f"{t}decoder.layers.{cfg.dec_layers - 1}."
    K = Wt.V4_DETECTORS
    for k in sd2:
        if k == t + "tgt_embed.weight":
            assert (sd4[k] - sd4[k][0:1]).abs().max() == 0 and not torch.equal(sd2[k], sd4[k])
        elif k in (p5 + "linear1.weight", p5 + "linear1.bias"):
            assert torch.equal(sd2[k][K:], sd4[k][K:]) and not torch.equal(sd2[k][:K], sd4[k][:K])
        elif k == p5 + "linear2.weight":
            assert torch.equal(sd2[k][:, K:], sd4[k][:, K:])
        else:
            assert torch.equal(sd2[k], sd4[k]), k
    with pytest.raises(ValueError):
        synthetic_state_dict(DTLRConfig.tiny(), 0, version=4)                       # calibrated for the Latin config only
    x = torch.stack(noise_lines(32, 128, 2048, seed=1000)[10:11])
    ob = OracleBatch(cfg, sd4, x, torch.zeros(1, 128, 2048, dtype=torch.bool), threads=16)
    lab, margin = query_decisions(ob.free["pred_logits"], ob.free["pred_boxes"], None)
    n_char = int((lab >= 0).sum())
    assert 20 <= n_char <= 200 and n_char == len(ob.strings[0])
    assert float(margin[lab >= 0].median()) > 1.0 and int((margin < 1e-2).sum()) <= 2
    idx = ob.debug["topk_idx"]
    perm = torch.randperm(cfg.num_queries, generator=torch.Generator().manual_seed(3))
    assert O.decode_blank(ob.teacher_forced(idx[:, perm])) == ob.strings           # any reordering of the SAME tokens: same string
    s = ob.self_sensitivity(1e-4)
    assert s["rank_slots_changed"] > 50 and s["strings_identical"] == "1/1", s


def _g9_reference_strings(g):
    """the REAL reference's blank-decoder strings (eps 0.003) from G9: per line the non-blank labels in reading order"""
    seq = g["ctc_argmax_eps003"].astype(np.int64)                  # [B, nq]: 0 = blank, c + 1 = class c, already in reading order
    return [[int(v) - 1 for v in row if v > 0] for row in seq]


def test_oracle_free_running_equals_the_real_reference_on_v4(golden_dir):
    """G9 (tests/golden/make_golden_v4.py): the REAL reference on generator-v4 weights, four lines of bench.py's batch, its OWN two-stage
    selection.  The oracle, free-running on the same lines, decodes the reference's strings exactly (4 / 4, ~70 characters each) -- although
    its selection is NOT the reference's list on any of the lines: the two CPU fp32 evaluations differ by 5e-6 in the two-stage scores and
    near-tied tokens trade ranks (the very effect that makes v2's rank-planted characters unusable for a free-running comparison)."""
    g = np.load(os.path.join(golden_dir, "g9_v4_free.npz"))
    assert int(g["generator_version"]) == 4
    cfg = DTLRConfig.latin()
    sd = synthetic_state_dict(cfg, 0, version=4)
    lines = noise_lines(32, 128, 2048, seed=1000)
    x = torch.stack([lines[int(r)] for r in g["rows"]])
    torch.set_num_threads(16)
    o = O.dino_forward(sd, cfg, x, mask=torch.zeros(len(x), 128, 2048, dtype=torch.bool), return_debug=True)
    serr = (o["_debug"]["topk_scores"] - torch.from_numpy(g["topk_scores"])).abs().max().item()
    assert serr < 1e-4, serr
    want = _g9_reference_strings(g)
    assert sum(len(w) for w in want) > 200
    assert O.decode_blank(o, 0.003) == want
    same_sel = (o["_debug"]["topk_idx"] == torch.from_numpy(g["topk_idx"].astype(np.int64))).all(1)
    print(f"[oracle vs reference, v4] score err {serr:.2e}; selection identical on {int(same_sel.sum())}/{len(x)} lines; strings identical on all")
    # teacher-forced on the reference's selection the outputs agree to fp32 rounding
    t = O.dino_forward(sd, cfg, None, forced_topk=torch.from_numpy(g["topk_idx"].astype(np.int64)), resume=o["_debug"])
    assert (t["pred_boxes"] - torch.from_numpy(g["pred_boxes"])).abs().max() < 2e-5
    idx = torch.from_numpy(g["top8_idx"].astype(np.int64))
    assert (torch.gather(t["pred_logits"], 2, idx) - torch.from_numpy(g["top8_val"])).abs().max() < 5e-3      # v4's detectors amplify ~1e3-fold
