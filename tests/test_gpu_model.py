"""End-to-end parity of the HIP/ROCm forward against the oracle and the reference goldens.  GPU only.

Discrete selections (two-stage top-k) amplify fp32 rounding into rank swaps of near-tied scores
between ANY two implementations (also between the reference on CPU and on CUDA), so parity is
checked in two parts: (1) the engine's selection is a valid top-k of the oracle's scores within a
tie tolerance, (2) with the selection pinned, logits agree within the north_star tolerance 1e-3.
"""
import os

import numpy as np
import pytest
import torch

from dtlr_amd import synth, weights
from dtlr_amd.config import DTLRConfig
from tests.util import selection_is_valid

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3     # BASELINE.json north_star: "logits within 1e-3 fp32"
BOX_TOL = 1e-4


def _model(cfg, sd, dtype=torch.float32):
    from dtlr_amd.dino import DINO
    m = DINO(cfg, compute_dtype=dtype)
    m.load_state_dict(sd)
    return m.eval().to("cuda:0")


def _cpu(out):
    return {"pred_logits": out["pred_logits"].float().cpu(), "pred_boxes": out["pred_boxes"].float().cpu()}


# the two parity-grade engines: exact-fp32 MFMA, and fp32 activations with split fp16 products (DTLREngine(split=True), round 4)
PARITY_ENGINES = [torch.float32, "f32s"]


@pytest.mark.parametrize("engine", PARITY_ENGINES, ids=["f32", "f32s"])
def test_tiny_model_vs_oracle_and_golden(golden_dir, engine):
    from oracle import dtlr_oracle as O
    g = np.load(os.path.join(golden_dir, "g2_tiny_model.npz"))
    cfg = DTLRConfig.tiny()
    sd = weights.synthetic_state_dict(cfg, 0)
    imgs = synth.stroke_lines(1, 32, 256, seed=5) + synth.noise_lines(1, 32, 192, seed=6)
    m = _model(cfg, sd, engine)
    m.return_aux = True
    out = m([i.cuda() for i in imgs], return_debug=True)
    d = out["_debug"]
    assert (d["memory"].cpu() - torch.from_numpy(g["memory"])).abs().max() < 2e-4
    assert (d["topk_scores"].cpu() - torch.from_numpy(g["topk_scores"])).abs().max() < 2e-4
    assert torch.equal(d["topk_idx"].cpu(), torch.from_numpy(g["topk_idx"]).long())     # no near ties at this size
    assert (out["pred_logits"].cpu() - torch.from_numpy(g["pred_logits"])).abs().max() < LOGIT_TOL
    assert (out["pred_boxes"].cpu() - torch.from_numpy(g["pred_boxes"])).abs().max() < BOX_TOL
    assert (out["interm_outputs"]["pred_logits"].cpu() - torch.from_numpy(g["interm_logits"])).abs().max() < LOGIT_TOL
    assert (out["interm_outputs"]["pred_boxes"].cpu() - torch.from_numpy(g["interm_boxes"])).abs().max() < BOX_TOL
    aux_l = torch.stack([a["pred_logits"] for a in out["aux_outputs"]]).cpu()
    assert (aux_l - torch.from_numpy(g["aux_logits"])).abs().max() < LOGIT_TOL
    ref = O.dino_forward(sd, cfg, imgs)
    assert (out["pred_logits"].cpu() - ref["pred_logits"]).abs().max() < LOGIT_TOL


def _reference_order_labels(g):
    """The reference's blank-decoder decision per QUERY (class index, -1 = blank) from the golden: `ctc_argmax_eps003` holds the
    argmax over [blank | classes] in reading order (queries sorted by box cx, dino.py:466-502)."""
    boxes = torch.from_numpy(g["pred_boxes"])
    seq = torch.from_numpy(g["ctc_argmax_eps003"].astype(np.int64))               # [B, nq], 0 = blank, c + 1 = class c
    order = torch.argsort(boxes[..., 0], dim=1, stable=True)
    lab = torch.empty_like(seq)
    lab.scatter_(1, order, seq - 1)
    return lab, order


@pytest.mark.parametrize("engine", PARITY_ENGINES, ids=["f32", "f32s"])
@pytest.mark.parametrize("tag", ["latin", "chinese"])
def test_full_model_vs_golden_and_oracle(golden_dir, tag, engine):
    """BASELINE configs: Latin 128x2048 and Chinese (C=7356) 128x2560, mixed-width pair (padding).  The parity-grade engines (exact
    fp32, and split fp16 products on fp32 activations) vs the REAL reference's stored outputs: scores, memory, logits, boxes, and the
    DECODED goldens (blank decoder decisions in reading order, PostProcess top-k, the NMS decoder's PostProcess call)."""
    from dtlr_amd import evaluation as E
    from dtlr_amd.dino import PostProcess
    from oracle import dtlr_oracle as O
    from tests.util import query_decisions
    g = np.load(os.path.join(golden_dir, f"g3_{tag}.npz"))
    assert int(g["generator_version"]) == weights.GENERATOR_VERSION
    cfg = DTLRConfig.latin() if tag == "latin" else DTLRConfig.chinese()
    sd = weights.synthetic_state_dict(cfg, 0)
    h, widths = int(g["height"]), [int(w) for w in g["widths"]]
    imgs = synth.stroke_lines(1, h, widths[0], seed=21) + synth.noise_lines(1, h, widths[1], seed=22)
    dimgs = [i.cuda() for i in imgs]
    m = _model(cfg, sd, engine)
    # (2) selection pinned to the reference's own -> compare with the reference's outputs
    ref_topk = torch.from_numpy(g["topk_idx"].astype(np.int64))
    out = m(dimgs, forced_topk=ref_topk.cuda(), return_debug=True)
    d = out["_debug"]
    score_err = (d["topk_scores"].cpu() - torch.from_numpy(g["topk_scores"])).abs().max().item()
    lerr = (torch.gather(out["pred_logits"].cpu(), 2, torch.from_numpy(g["top8_idx"].astype(np.int64))) - torch.from_numpy(g["top8_val"])).abs().max().item()
    berr = (out["pred_boxes"].cpu() - torch.from_numpy(g["pred_boxes"])).abs().max().item()
    print(f"[{engine} vs reference golden, {tag}] score err {score_err:.2e}, logit err {lerr:.2e}, box err {berr:.2e}")
    assert score_err < 5e-4, score_err
    assert (d["memory"][:, ::67].cpu() - torch.from_numpy(g["memory_rows"])).abs().max() < 5e-4
    idx = torch.from_numpy(g["top8_idx"].astype(np.int64))
    assert (torch.gather(out["pred_logits"].cpu(), 2, idx) - torch.from_numpy(g["top8_val"])).abs().max() < LOGIT_TOL
    assert (out["pred_boxes"].cpu() - torch.from_numpy(g["pred_boxes"])).abs().max() < BOX_TOL
    # ---- decoded goldens.  Blank decoder (eps 0.003): per-query decision identical to the reference's wherever the decision
    # margin exceeds the logit tolerance, and the reading order identical except among queries whose cx differ by < 2 * BOX_TOL
    host = _cpu(out)
    ref_lab, ref_order = _reference_order_labels(g)
    lab, margin = query_decisions(host["pred_logits"], host["pred_boxes"], 0.003)
    safe = margin > 2 * LOGIT_TOL
    assert safe.float().mean() > 0.98, safe.float().mean()
    assert torch.equal(lab[safe], ref_lab[safe])
    cx_ref = torch.from_numpy(g["pred_boxes"])[..., 0]
    order = torch.argsort(host["pred_boxes"][..., 0], dim=1, stable=True)
    moved = order != ref_order
    for b in range(order.shape[0]):                        # a query may only change rank inside a run of near-equal cx
        pos = torch.nonzero(moved[b]).flatten()
        if len(pos):
            assert (cx_ref[b, order[b, pos]] - cx_ref[b, ref_order[b, pos]]).abs().max() < 2 * BOX_TOL
    # the product decoder on the engine's outputs == the reference's decoded sequence where all of it is safe
    dec = E.decode_blank(out, 0.003)
    for b in range(len(dec)):
        want = [int(x) for x in ref_lab[b, ref_order[b]] if x >= 0]
        if bool(safe[b].all()) and not bool(moved[b].any()):
            assert dec[b] == want
        assert abs(len(dec[b]) - len(want)) <= int((~safe[b]).sum())
    # PostProcess (num_select = 300): the sorted score vector is stable under perturbation; labels / boxes must match wherever a
    # score is separated from its neighbours by more than the score tolerance (sigmoid slope <= 1/4)
    B = host["pred_logits"].shape[0]
    pp = PostProcess(num_select=cfg.num_select)(out, torch.ones(B, 2).cuda())
    sc = torch.stack([p["scores"] for p in pp]).cpu()
    gs = torch.from_numpy(g["pp_scores"])
    assert (sc - gs).abs().max() < LOGIT_TOL / 4 + 1e-6
    sep = torch.ones_like(gs, dtype=torch.bool)
    close = (gs[:, :-1] - gs[:, 1:]) < 2 * (LOGIT_TOL / 4)
    sep[:, :-1] &= ~close
    sep[:, 1:] &= ~close
    sep[:, -1] = False                                      # the cut itself: the next score is not stored
    assert sep.float().mean() > 0.2
    assert torch.equal(torch.stack([p["labels"] for p in pp]).cpu()[sep], torch.from_numpy(g["pp_labels"]).long()[sep])
    assert (torch.stack([p["boxes"] for p in pp]).cpu()[sep] - torch.from_numpy(g["pp_boxes"])[sep]).abs().max() < 2 * BOX_TOL
    # the NMS decoder's PostProcess call (num_select = 900, IoU 0.5) per sample: kept (score, label) pairs above TH = 0.3
    post = PostProcess(num_select=cfg.num_queries, nms_iou_threshold=0.5)
    for b in range(B):
        one = {"pred_logits": out["pred_logits"][b:b + 1], "pred_boxes": out["pred_boxes"][b:b + 1]}
        r = post(one, torch.tensor([[1.0, 1.0]]).cuda())[0]
        rs, rl = torch.from_numpy(g[f"nms_scores_{b}"]), torch.from_numpy(g[f"nms_labels_{b}"]).long()
        conf_ref = [(int(l)) for s_, l in zip(rs.tolist(), rl.tolist()) if s_ > 0.3 + 1e-3]
        conf_got = [(int(l)) for s_, l in zip(r["scores"].cpu().tolist(), r["labels"].cpu().tolist()) if s_ > 0.3 + 1e-3]
        assert sorted(conf_ref) == sorted(conf_got)
    # (1) free-running selection is a valid top-k of the reference's scores (tie-aware; tolerance = the measured score error)
    free = m(dimgs, return_debug=True)
    fidx = free["_debug"]["topk_idx"].cpu()
    assert selection_is_valid(fidx, torch.from_numpy(g["topk_scores"]), cfg.num_queries, tol=2 * score_err + 1e-7)
    # ... and with the oracle following that selection, logits agree
    ref = O.dino_forward(sd, cfg, imgs, forced_topk=fidx)
    assert (free["pred_logits"].cpu() - ref["pred_logits"]).abs().max() < LOGIT_TOL
    assert (free["pred_boxes"].cpu() - ref["pred_boxes"]).abs().max() < BOX_TOL
    # FREE-RUNNING strings (round 5): the engine with its own selection against the REFERENCE's stored decode (its own selection), no
    # tolerance, no safe set: wherever the two selections are the same list the strings must simply be equal.  (Where they are not, two
    # near-tied tokens traded ranks -- validated above -- and with them their tgt_embed rows: deformable_transformer.py:354-355.)
    fdec = E.decode_blank(free, 0.003)
    n_same_sel = n_same_str = 0
    for b in range(len(fdec)):
        want = [int(x) for x in ref_lab[b, ref_order[b]] if x >= 0]
        n_same_str += int(fdec[b] == want)
        if torch.equal(fidx[b], ref_topk[b]):
            n_same_sel += 1
            assert fdec[b] == want, (b, fdec[b], want)
    print(f"[{engine} free-running vs reference golden, {tag}] selection identical on {n_same_sel}/{len(fdec)} lines, strings identical on {n_same_str}/{len(fdec)}")


# Bounds of the bf16 (bench) engine against the fp32 CPU oracle on the same selection (measured on MI355X, margin-bearing
# generator v2): max |logit difference| and max |box difference|.  REGRESSION ALARMS, not the north_star's 1e-3: only the
# f32 / f32s engines are held to that (test_parity_engines_bench_batch_vs_oracle).
BF16_LOGIT_BOUND = {"latin": 0.3, "chinese": 0.4}      # measured: 0.126 (Latin pair), mean 0.017
BF16_BOX_BOUND = 2e-2                                   # measured: 8.4e-3 max over (cx, cy, w, h), mean 8e-4
# ... and of the fp16 build of the same kernels (libdtlr_hip_f16.so; round 3): 8x finer rounding everywhere
F16_LOGIT_BOUND = {"latin": 0.06, "chinese": 0.08}     # measured: 0.024 (8 lines of the bench batch), mean 0.0022
F16_BOX_BOUND = 4e-3                                    # measured: 1.4e-3
HALF = {"bf16": torch.bfloat16, "f16": torch.float16}


def _bounds(half, tag):
    return (BF16_LOGIT_BOUND[tag], BF16_BOX_BOUND) if half == "bf16" else (F16_LOGIT_BOUND[tag], F16_BOX_BOUND)


@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("tag", ["latin", "chinese"])
def test_bf16_engine_vs_oracle_decoded_strings(golden_dir, tag, half):
    """The BENCHED engine (bf16 operands) against O.dino_forward -- not against the fp32 HIP engine -- on the BASELINE shapes
    (Latin 128x2048, Chinese C = 7356 128x2560; mixed-width pair): a stated logit bound, labels identical on every query whose
    oracle decision margin exceeds the measured bf16 logit error, reading-order strings identical on those (CER == 0), for both
    decoders' eps."""
    from dtlr_amd import evaluation as E_
    from oracle import dtlr_oracle as O
    from tests.util import compare_decoded
    g = np.load(os.path.join(golden_dir, f"g3_{tag}.npz"))
    cfg = DTLRConfig.latin() if tag == "latin" else DTLRConfig.chinese()
    sd = weights.synthetic_state_dict(cfg, 0)
    h, widths = int(g["height"]), [int(w) for w in g["widths"]]
    imgs = synth.stroke_lines(1, h, widths[0], seed=21) + synth.noise_lines(1, h, widths[1], seed=22)
    ref = O.dino_forward(sd, cfg, imgs, return_debug=True)
    idx = ref["_debug"]["topk_idx"]
    m16 = _model(cfg, sd, HALF[half])
    o16 = m16([i.cuda() for i in imgs], forced_topk=idx.cuda(), return_debug=True)
    got = _cpu(o16)
    err = (got["pred_logits"] - ref["pred_logits"]).abs()
    berr = (got["pred_boxes"] - ref["pred_boxes"]).abs()
    E, Eb, Ecx = err.max().item(), berr.max().item(), berr[..., 0].max().item()
    print(f"[{half} vs oracle, {tag}] logit err max {E:.4f} mean {err.mean().item():.5f}; box err max {Eb:.5f} (cx {Ecx:.5f}) mean {berr.mean().item():.6f}")
    lb, bb = _bounds(half, tag)
    assert E < lb and Eb < bb, (E, Eb)
    for eps in (None, 0.003):
        st = compare_decoded(ref["pred_logits"], ref["pred_boxes"], got["pred_logits"], got["pred_boxes"], eps, E, Ecx)   # reading order depends on cx only
        print(f"[bf16 vs oracle, {tag}, eps={eps}] {st}")
        assert st["label_mismatch_on_safe"] == 0 and st["strings_equal"], st      # CER(bf16 vs oracle) == 0 on the safe queries
        assert st["safe_frac"] > 0.9 and st["raw_label_agree"] > 0.99, st               # the gate covers most queries
        if half == "f16":                                                               # fp16: the gate covers (nearly) ALL characters
            assert st["safe_frac"] > 0.995 and st["safe_chars"] >= 0.5 * st["chars_ref"], st
        # unrestricted strings (every query, including the unsafe ones): reported, and bounded -- bf16 rounding may flip a query
        # whose margin is below the logit error or swap two characters whose cx differ by less than the box error, nothing else
        a, b = O.decode_blank(ref, eps), E_.decode_blank(o16, eps)
        dist = sum(O.levenshtein(x, y) for x, y in zip(a, b))
        n = sum(len(x) for x in a)
        print(f"[bf16 vs oracle, {tag}, eps={eps}] unrestricted CER {dist}/{n} = {dist / max(n, 1):.4f}")
        assert dist <= 2 * (st["chars_ref"] - st["safe_chars"]) + 2 * int((1 - st["safe_frac"]) * ref["pred_logits"].shape[0] * ref["pred_logits"].shape[1])
    # the bf16 engine's own selection is a valid top-k of the oracle's scores within the measured score error
    free = m16([i.cuda() for i in imgs], return_debug=True)
    serr = (free["_debug"]["topk_scores"].cpu() - ref["_debug"]["topk_scores"]).abs().max().item()
    print(f"[{half} vs oracle, {tag}] two-stage score err {serr:.2e}")
    assert serr < (0.1 if half == "bf16" else 0.02)             # measured 0.034 (bf16 memory, fp32 scores) / 0.005 (fp16)
    assert selection_is_valid(free["_debug"]["topk_idx"].cpu(), ref["_debug"]["topk_scores"], cfg.num_queries, tol=2 * serr + 1e-7)


def test_decoders_and_cer_identical_to_oracle():
    """Same logits in -> identical label sequences / CER out, for both decoders (evaluation.py:94-158),
    both blank eps (0.03/C and 0.003) and PostProcess (dino.py:985-1046)."""
    from dtlr_amd import evaluation as E
    from dtlr_amd.dino import PostProcess
    from oracle import dtlr_oracle as O
    cfg = DTLRConfig.tiny(num_classes=23)
    sd = weights.synthetic_state_dict(cfg, 3)
    imgs = synth.stroke_lines(3, 32, [256, 160, 224], seed=9)
    m = _model(cfg, sd)
    out = m([i.cuda() for i in imgs])
    host = _cpu(out)
    for eps in (None, 0.003):
        assert E.decode_blank(out, eps) == O.decode_blank(host, eps)
        assert (E.blank_probabilities(out, 0.003).cpu() - O.blank_probabilities(host, 0.003)).abs().max() < 1e-6
    for th, nm in ((0.3, 0.5), (0.05, 0.3), (0.01, 0.7)):
        assert E.decode_nms(out, PostProcess(), th, nm) == O.decode_nms(host, th, nm)
    pp = PostProcess(num_select=cfg.num_select)(out, torch.ones(3, 2).cuda())
    po = O.post_process(host, torch.ones(3, 2), cfg.num_select)
    for a, b in zip(pp, po):
        assert torch.equal(a["labels"].cpu(), b["labels"])
        assert (a["scores"].cpu() - b["scores"]).abs().max() < 1e-6
        assert (a["boxes"].cpu() - b["boxes"]).abs().max() < 1e-6
    gt = [[1, 2, 3, 4], [5, 6], [7, 8, 9]]
    pred = E.decode_blank(out)
    for p, t in zip(pred, gt):
        assert E.character_error_rate(p, t) == O.character_error_rate_engine(p, t)


@pytest.mark.parametrize("engine", PARITY_ENGINES, ids=["f32", "f32s"])
def test_full_size_batch_properties(engine):
    """BASELINE bs=32 at 128x2048 (fp32): per-line independence -- a line's result does not depend on
    its batch neighbours (no cross-sample op anywhere in DINO.forward) -- and padding equivalence:
    a narrower line padded into the 2048 canvas gives the same logits as when run alone."""
    cfg = DTLRConfig.latin()
    sd = weights.synthetic_state_dict(cfg, 0)
    m = _model(cfg, sd, engine)
    imgs = [i.cuda() for i in synth.noise_lines(32, 128, 2048, seed=4)]
    full = m(imgs, return_debug=True)
    idx = full["_debug"]["topk_idx"]
    sub = m(imgs[7:9], forced_topk=idx[7:9])
    assert (sub["pred_logits"] - full["pred_logits"][7:9]).abs().max() < LOGIT_TOL
    assert (sub["pred_boxes"] - full["pred_boxes"][7:9]).abs().max() < BOX_TOL
    assert torch.isfinite(full["pred_logits"]).all() and torch.isfinite(full["pred_boxes"]).all()
    assert (full["pred_boxes"] >= 0).all() and (full["pred_boxes"] <= 1).all()


BENCH_PARITY_ROWS = [0, 10, 21, 31]          # four of the 16 rows bench.py's parity leg uses (--parity-lines 16 over 32 lines)
_ORACLE_BATCH = {}


def _bench_oracle_batch():
    """the CPU oracle's free run of BENCH_PARITY_ROWS of bench.py's batch (seed 1000), once per session"""
    from oracle.parity import OracleBatch
    if "ob" not in _ORACLE_BATCH:
        cfg = DTLRConfig.latin()
        sd = weights.synthetic_state_dict(cfg, 0)
        imgs = synth.noise_lines(32, 128, 2048, seed=1000)
        x = torch.stack([imgs[r] for r in BENCH_PARITY_ROWS])
        _ORACLE_BATCH["ob"] = (OracleBatch(cfg, sd, x, torch.zeros(x.shape[0], 128, 2048, dtype=torch.bool)), cfg, sd, imgs)
    return _ORACLE_BATCH["ob"]


@pytest.mark.parametrize("engine", PARITY_ENGINES, ids=["f32", "f32s"])
def test_parity_engines_bench_batch_vs_oracle(engine):
    """BASELINE configs[1]'s batch exactly as bench.py runs it (32 unpadded 128x2048 lines, seed 1000), parity-grade engines, FULL batch
    size, against the CPU oracle on four of its lines (oracle/parity.py, the checker bench.py prints):
      teacher-forced  logits within north_star's 1e-3, boxes within 1e-4, and the decoded strings IDENTICAL (CER 0 over all 900 queries);
      free-running    the engine's own selection vs the oracle's own: lines with the same selection have identical strings, unconditionally;
                      the count of identical lines is printed next to the oracle's sensitivity to its own score noise."""
    from dtlr_amd import evaluation as E
    ob, cfg, sd, imgs = _bench_oracle_batch()
    m = _model(cfg, sd, engine)
    out = m(torch.stack(imgs).cuda(), return_debug=True)
    d = out["_debug"]
    name = "f32s" if engine == "f32s" else "f32"
    rows = BENCH_PARITY_ROWS
    r = ob.compare(name, out["pred_logits"][rows], out["pred_boxes"][rows], d["topk_idx"][rows], d["topk_scores"][rows])
    print(f"[{name} bs32 vs oracle] {r}")
    tf, fr = r["teacher_forced"], r["free_running"]
    assert tf["logit_err_max"] < LOGIT_TOL and tf["box_err_max"] < BOX_TOL, tf
    assert tf["edit_distance"] == 0 and tf["label_flips"] == 0, tf               # identical strings on the same selection: no accounting
    assert r["parity_gate"]
    k, n = (int(v) for v in fr["strings_identical_given_identical_selection"].split("/"))
    assert k == n, fr                                                             # same selection -> same strings, free-running
    assert fr["two_stage_score_err_max"] < 1e-4, fr
    # the product decoder == the oracle's decoder on the engine's outputs
    sub = {"pred_logits": out["pred_logits"][rows], "pred_boxes": out["pred_boxes"][rows]}
    assert E.decode_blank(sub) == O_decode(sub)


_ORACLE_BATCH_V4 = {}


def _bench_oracle_batch_v4():
    """generator-v4 weights (rank-invariant content queries, image-driven characters: dtlr_amd/weights.py) on BENCH_PARITY_ROWS of the bench batch"""
    from oracle.parity import OracleBatch
    if "ob" not in _ORACLE_BATCH_V4:
        cfg = DTLRConfig.latin()
        sd = weights.synthetic_state_dict(cfg, 0, version=4)
        imgs = synth.noise_lines(32, 128, 2048, seed=1000)
        x = torch.stack([imgs[r] for r in BENCH_PARITY_ROWS])
        _ORACLE_BATCH_V4["ob"] = (OracleBatch(cfg, sd, x, torch.zeros(x.shape[0], 128, 2048, dtype=torch.bool)), cfg, sd, imgs)
    return _ORACLE_BATCH_V4["ob"]


# free-running CER bounds of the 16-bit engines on generator v4 (measured on MI355X in round 5; what differs is mostly the SET of selected
# tokens at the 900-th score -- the oracle against itself at the same score error shows the same CER).  These are REGRESSION ALARMS set from
# measurements, not parity statements: the 16-bit engines are throughput engines that do not reproduce the reference's strings (README /
# DESIGN section 2); the parity statement -- identical strings, no tolerance -- is asserted for f32 and f32s only.
V4_FREE_CER_BOUND = {"bf16": 0.35, "f16": 0.10}      # measured: 0.264 / 0.035 on these four lines (0.269 / 0.062 on the bench's eight)


@pytest.mark.parametrize("engine", ["f32", "f32s", "f16", "bf16"])
def test_free_running_strings_vs_oracle_on_image_driven_weights(engine):
    """FREE-RUNNING, no teacher forcing, no tolerance: the engine with its own two-stage selection against the CPU oracle with ITS own, on
    the bench batch (32 lines, 128x2048) with generator-v4 weights -- characters read from the image, identical content queries, so a rank
    swap among near-tied tokens permutes queries without changing the decode (the property a trained recogniser has and v2's planted
    characters lack).  fp32-grade engines: the decoded strings of the compared lines are IDENTICAL to the oracle's.  16-bit engines: CER
    bounded (their score error moves ~50 of the 900 selected tokens per line across the cut)."""
    ob, cfg, sd, imgs = _bench_oracle_batch_v4()
    dt = {"f32": torch.float32, "f32s": "f32s", "f16": torch.float16, "bf16": torch.bfloat16}[engine]
    m = _model(cfg, sd, dt)
    out = m(torch.stack(imgs).cuda(), return_debug=True)
    d = out["_debug"]
    rows = BENCH_PARITY_ROWS
    r = ob.compare(engine, out["pred_logits"][rows], out["pred_boxes"][rows], d["topk_idx"][rows], d["topk_scores"][rows], budgeted=False)
    print(f"[{engine} free-running on v4 weights] {r}")
    fr = r["free_running"]
    assert sum(len(s_) for s_ in ob.strings) > 100                 # the lines do carry text
    if engine in ("f32", "f32s"):
        assert fr["edit_distance"] == 0 and fr["strings_identical_free_running"] == f"{len(rows)}/{len(rows)}", fr
    else:
        assert fr["cer_free_running"] < V4_FREE_CER_BOUND[engine], fr


@pytest.mark.parametrize("engine", PARITY_ENGINES, ids=["f32", "f32s"])
def test_free_running_strings_equal_the_real_reference_on_v4(golden_dir, engine):
    """The statement of the round-4 review, literally: FREE-RUNNING engine strings against FREE-RUNNING REFERENCE strings.  G9 holds what the
    real reference (tests/golden/make_golden_v4.py) decodes for four lines of bench.py's batch on generator-v4 weights with its own two-stage
    selection; the parity-grade engines, with THEIR own selection, must decode exactly that (blank decoder, eps 0.003: dino.py:466-502)."""
    from dtlr_amd import evaluation as E
    g = np.load(os.path.join(golden_dir, "g9_v4_free.npz"))
    cfg = DTLRConfig.latin()
    sd = weights.synthetic_state_dict(cfg, 0, version=4)
    lines = synth.noise_lines(32, 128, 2048, seed=1000)
    m = _model(cfg, sd, engine)
    out = m(torch.stack([lines[int(r)] for r in g["rows"]]).cuda(), return_debug=True)
    seq = g["ctc_argmax_eps003"].astype(np.int64)
    want = [[int(v) - 1 for v in row if v > 0] for row in seq]
    got = E.decode_blank(out, 0.003)
    same_sel = (out["_debug"]["topk_idx"].cpu() == torch.from_numpy(g["topk_idx"].astype(np.int64))).all(1)
    print(f"[{engine} free-running vs the REAL reference, v4] selection identical on {int(same_sel.sum())}/4 lines; strings identical on "
          f"{sum(int(a == b) for a, b in zip(got, want))}/4; characters {[len(w) for w in want]}")
    assert got == want


def O_decode(sub):
    from oracle import dtlr_oracle as O
    return O.decode_blank({k: v.float().cpu() for k, v in sub.items()})


def _bf16_vs_oracle(tag, cfg, sd, imgs, o16, rows, eps_list=(None,), half="bf16"):
    """Shared gate: rows `rows` of a bf16 engine output against O.dino_forward on the same lines with the same selection."""
    from oracle import dtlr_oracle as O
    from tests.util import compare_decoded
    idx = o16["_debug"]["topk_idx"][rows].cpu()
    ref = O.dino_forward(sd, cfg, [imgs[r] for r in rows], forced_topk=idx)
    got = {"pred_logits": o16["pred_logits"][rows].float().cpu(), "pred_boxes": o16["pred_boxes"][rows].float().cpu()}
    E = (got["pred_logits"] - ref["pred_logits"]).abs().max().item()
    Eb = (got["pred_boxes"] - ref["pred_boxes"]).abs().max().item()
    Ecx = (got["pred_boxes"][..., 0] - ref["pred_boxes"][..., 0]).abs().max().item()
    print(f"[{half} vs oracle, {tag}] logit err max {E:.4f}, box err max {Eb:.5f} (cx {Ecx:.5f})")
    lb, bb = _bounds(half, "chinese" if cfg.num_classes > 1000 else "latin")
    assert E < lb and Eb < bb, (E, Eb)
    for eps in eps_list:
        st = compare_decoded(ref["pred_logits"], ref["pred_boxes"], got["pred_logits"], got["pred_boxes"], eps, E, Ecx)
        print(f"[bf16 vs oracle, {tag}, eps={eps}] {st}")
        assert st["label_mismatch_on_safe"] == 0 and st["strings_equal"] and st["safe_frac"] > 0.9, st


@pytest.mark.parametrize("half", ["bf16", "f16"])
def test_bf16_bench_batch_vs_oracle_and_line_independence(half):
    """BASELINE configs[1] exactly as bench.py runs it: bf16 engine, 32 unpadded 128x2048 lines, its own selection.  (a) lines
    0 and 17 of the batch against the CPU oracle following the engine's selection: stated logit bound, decoded strings identical
    on the safe queries; (b) per-line independence in bf16: a line's result does not depend on its batch neighbours (the
    sub-batch runs other tile shapes / kernel variants, so the comparison uses the bf16 bound, and the decoded labels of the safe
    queries must be identical)."""
    from tests.util import compare_decoded
    cfg = DTLRConfig.latin()
    sd = weights.synthetic_state_dict(cfg, 0)
    imgs = synth.noise_lines(32, 128, 2048, seed=4)
    m16 = _model(cfg, sd, HALF[half])
    full = m16(torch.stack(imgs).cuda(), return_debug=True)
    assert torch.isfinite(full["pred_logits"]).all() and torch.isfinite(full["pred_boxes"]).all()
    assert (full["pred_boxes"] >= 0).all() and (full["pred_boxes"] <= 1).all()
    _bf16_vs_oracle("bench batch rows 0,17", cfg, sd, imgs, full, [0, 17], eps_list=(None, 0.003), half=half)
    sub = m16([imgs[7].cuda(), imgs[8].cuda()], forced_topk=full["_debug"]["topk_idx"][7:9])
    d = (sub["pred_logits"].float() - full["pred_logits"][7:9].float()).abs().max().item()
    db = (sub["pred_boxes"].float() - full["pred_boxes"][7:9].float()).abs().max().item()
    print(f"[{half} line independence] logit diff {d:.4f}, box diff {db:.5f}")
    assert d < _bounds(half, "latin")[0] and db < _bounds(half, "latin")[1]
    st = compare_decoded(full["pred_logits"][7:9].float().cpu(), full["pred_boxes"][7:9].float().cpu(),
                         sub["pred_logits"].float().cpu(), sub["pred_boxes"].float().cpu(), None, max(d, 1e-3),
                         max((sub["pred_boxes"][..., 0].float() - full["pred_boxes"][7:9, :, 0].float()).abs().max().item(), 1e-5))
    assert st["label_mismatch_on_safe"] == 0 and st["strings_equal"], st


@pytest.mark.parametrize("half", ["bf16", "f16"])
def test_bf16_padded_batch_vs_oracle(half):
    """bf16 engine on a MIXED-WIDTH (zero-padded, masked) batch: exercises the padding-row epilogue of the value GEMMs, the
    batched decoder value projection, the fused FFN and the LDS MSDA kernel on padded maps -- against the CPU oracle."""
    cfg = DTLRConfig.latin()
    sd = weights.synthetic_state_dict(cfg, 0)
    imgs = synth.stroke_lines(3, 128, [2048, 1536, 1792], seed=17)
    m16 = _model(cfg, sd, HALF[half])
    o16 = m16([i.cuda() for i in imgs], return_debug=True)
    assert torch.isfinite(o16["pred_logits"]).all() and torch.isfinite(o16["pred_boxes"]).all()
    _bf16_vs_oracle("padded 2048/1536/1792", cfg, sd, imgs, o16, [0, 1, 2], half=half)


@pytest.mark.parametrize("half", ["bf16", "f16"])
def test_bf16_chinese_cfg5_batch_properties(half):
    """BASELINE configs[4]: Chinese head (C = 7356), 32 mixed-length lines (widths seeded from {1536..2560}) padded to 128x2560,
    bf16 engine: finite outputs, boxes inside the canvas, the padded part of a line never selected by the two-stage top-k, line
    independence on the same canvas, and two lines against the CPU oracle."""
    from tests.util import compare_decoded
    cfg = DTLRConfig.chinese()
    sd = weights.synthetic_state_dict(cfg, 0)
    widths = synth.mixed_widths(32, [1536, 1792, 2048, 2304, 2560], seed=5)
    widths[0] = 2560                                               # the canvas width is pinned by line 0
    imgs = synth.noise_lines(32, 128, widths, seed=44)
    m16 = _model(cfg, sd, HALF[half])
    full = m16([i.cuda() for i in imgs], return_debug=True)
    assert tuple(full["pred_logits"].shape) == (32, cfg.num_queries, 7356)
    assert torch.isfinite(full["pred_logits"]).all() and torch.isfinite(full["pred_boxes"]).all()
    assert (full["pred_boxes"] >= 0).all() and (full["pred_boxes"] <= 1).all()
    g = full["_debug"]["geometry"]
    picked_pad = torch.gather(g["mask_flat"], 1, full["_debug"]["topk_idx"])
    assert not picked_pad.any()                                    # padded tokens score the bare bias: never in the top 900
    rows = [3, 20]
    _bf16_vs_oracle("cfg5 rows 3,20", cfg, sd, [imgs[0]] + [imgs[r] for r in rows],
                    {"pred_logits": full["pred_logits"][[0] + rows], "pred_boxes": full["pred_boxes"][[0] + rows],
                     "_debug": {"topk_idx": full["_debug"]["topk_idx"][[0] + rows]}}, [0, 1, 2], half=half)   # line 0 pins the 2560 canvas
    sub = m16([imgs[0].cuda(), imgs[3].cuda(), imgs[20].cuda()], forced_topk=full["_debug"]["topk_idx"][[0, 3, 20]])
    d = (sub["pred_logits"].float() - full["pred_logits"][[0, 3, 20]].float()).abs().max().item()
    assert d < _bounds(half, "chinese")[0], d
    st = compare_decoded(full["pred_logits"][[0, 3, 20]].float().cpu(), full["pred_boxes"][[0, 3, 20]].float().cpu(),
                         sub["pred_logits"].float().cpu(), sub["pred_boxes"].float().cpu(), None, max(d, 1e-3), _bounds(half, "chinese")[1])
    assert st["label_mismatch_on_safe"] == 0, st


@pytest.mark.parametrize("half", ["bf16", "f16"])
def test_chinese_heads_token_stationary_equals_tiled(half):
    """Round 5: the Chinese model's three 7356-class heads (two-stage scores, interm logits, final logits) on the token-stationary kernel
    (dtlr_head_ts; DTLREngine.head_ts_min_classes = 1024) against the tiled-GEMM forms they replace (head_ts_min_classes beyond the class
    count): the same split products in another summation order -- scores / logits equal to fp32 rounding, and with the selection pinned
    the decoded strings identical."""
    from dtlr_amd import evaluation as E_
    from dtlr_amd.engine import DTLREngine
    cfg = DTLRConfig.chinese()
    sd = weights.synthetic_state_dict(cfg, 0)
    x = torch.stack(synth.noise_lines(3, 128, 1536, seed=61)).cuda()
    mask = torch.zeros((3, 128, 1536), dtype=torch.bool, device="cuda:0")
    eng = DTLREngine(cfg, sd, "cuda:0", HALF[half])
    assert eng.head_ts_min_classes <= cfg.num_classes
    new = eng.forward(x, mask, has_padding=False, return_debug=True)
    eng.head_ts_min_classes, eng.head_ts_scores = 10 ** 9, False
    old = eng.forward(x, mask, has_padding=False, return_debug=True, forced_topk=new["_debug"]["topk_idx"])
    free_old = eng.forward(x, mask, has_padding=False, return_debug=True)
    ds = (new["_debug"]["topk_scores"] - free_old["_debug"]["topk_scores"]).abs().max().item()
    dl = (new["pred_logits"] - old["pred_logits"]).abs().max().item()
    di = (new["interm_outputs"]["pred_logits"] - old["interm_outputs"]["pred_logits"]).abs().max().item()
    print(f"[{half} chinese heads: token-stationary vs tiled] score diff {ds:.2e}, logit diff {dl:.2e}, interm diff {di:.2e}")
    assert ds < 2e-5 and dl < 5e-5 and di < 5e-5
    assert E_.decode_blank(new) == E_.decode_blank(old)


def test_head_resize_checkpoint_flow_forward(tmp_path):
    """a model built from the stock config (23 classes here) ingests a checkpoint whose heads were
    rebuilt to another charset (11 classes) through evaluation.load_model (evaluation.py:51-88) and then produces the
    oracle's logits for that state dict: the engine sizes its class heads from the weights, not from the config."""
    import dataclasses
    from dtlr_amd import evaluation as E
    from dtlr_amd.dino import DINO
    from oracle import dtlr_oracle as O
    cfg = DTLRConfig.tiny()
    cfg11 = dataclasses.replace(cfg, num_classes=11)            # label_enc keeps its size: no --new_label_enc
    sd = weights.synthetic_state_dict(cfg11, 4)
    ck = {k: v for k, v in sd.items() if not k.startswith("transformer.decoder.class_embed.")}
    ck["transformer.decoder.class_embed.weight"] = torch.zeros(11, 256)
    ck["transformer.decoder.class_embed.bias"] = torch.zeros(11)
    torch.save({"model": ck}, tmp_path / "checkpoint.pth")
    m = E.load_model(DINO(cfg), str(tmp_path / "checkpoint.pth"), device="cuda:0", new_class_embedding=True, charset_size=11)
    imgs = synth.stroke_lines(2, 32, 256, seed=9)
    out = m([i.cuda() for i in imgs], return_debug=True)
    assert tuple(out["pred_logits"].shape) == (2, cfg.num_queries, 11)
    ref = O.dino_forward(sd, cfg11, imgs, forced_topk=out["_debug"]["topk_idx"].cpu())
    assert (out["pred_logits"].cpu() - ref["pred_logits"]).abs().max() < LOGIT_TOL
    assert (out["pred_boxes"].cpu() - ref["pred_boxes"]).abs().max() < BOX_TOL


def test_evaluation_cli_on_synthetic_assets(tmp_path):
    """`python -m dtlr_amd.evaluation` end to end on synthetic assets (checkpoint.pth + a folder of line images + labels): the
    harness of evaluation.py:460-659 -- preprocess -> forward -> decode -> CER/WER -> the reference's output files.  With
    `--batching exact` the predictions equal the oracle run the reference's way, ONE image at a time."""
    import json
    from PIL import Image
    from dtlr_amd import eval_harness as H
    from oracle import dtlr_oracle as O
    from tests.util import preproc_image
    cs = H.load_charset(None)
    cfg = DTLRConfig.tiny(num_classes=len(cs))
    sd = weights.synthetic_state_dict(cfg, 6)
    torch.save({"model": sd, "epoch": 3}, tmp_path / "checkpoint.pth")
    img_dir = tmp_path / "lines"
    img_dir.mkdir()
    shapes = [(40, 300), (40, 300), (33, 410), (40, 300), (25, 160)]
    texts = ["hello world", "The B B C , 1, 2", "x - y", "abc def", "q"]
    imgs = []
    for k, (h, w) in enumerate(shapes):
        im = preproc_image(h, w, 20 + k)
        imgs.append(im)
        Image.fromarray(im, "RGB").save(img_dir / f"l{k:02d}.png")
    (tmp_path / "labels.json").write_text(json.dumps([[f"l{k:02d}", t] for k, t in enumerate(texts)]))
    res = H.main(["--config", "tiny", "--weights", str(tmp_path / "checkpoint.pth"), "--images", str(img_dir),
                  "--labels", str(tmp_path / "labels.json"), "--dataset", "IAM", "--out", str(tmp_path / "stats"),
                  "--dtype", "f32", "--batch", "3", "--size", "32", "--max_size", "256"])
    # the oracle, one image at a time (evaluation.py:499), same transform parameters
    want = []
    for im in imgs:
        x, m = O.preprocess_lines([im], size=32, max_size=256)
        want.append(O.decode_blank(O.dino_forward(sd, cfg, x, mask=m))[0])
    got_str = res["list_preds_str"]
    assert got_str == ["".join(cs[i] for i in p) for p in want]
    ref = H.evaluate_predictions(want, texts, cs, "IAM", "default")
    assert res["CER_list"] == ref["CER_list"] and res["WER_list"] == ref["WER_list"]
    d = tmp_path / "stats" / "IAM"
    assert sorted(os.listdir(d)) == ["cer_TH_None_NMS_None.txt", "cer_list.npy", "dict_char.json", "list_gt.txt", "list_preds.txt"]
    assert (d / "list_gt.txt").read_text().splitlines() == texts
    # per-sample error isolation (evaluation.py:498-504): an unreadable image is reported and skipped, the other lines are unaffected
    (img_dir / "l05.png").write_bytes(b"not a png at all")
    (tmp_path / "labels6.json").write_text(json.dumps([[f"l{k:02d}", t] for k, t in enumerate(texts + ["lost line"])]))
    res6 = H.main(["--config", "tiny", "--weights", str(tmp_path / "checkpoint.pth"), "--images", str(img_dir),
                   "--labels", str(tmp_path / "labels6.json"), "--dataset", "IAM", "--out", str(tmp_path / "stats6"),
                   "--dtype", "f32", "--batch", "3", "--size", "32", "--max_size", "256"])
    assert res6["list_preds_str"] == got_str and res6["list_gt_str"] == res["list_gt_str"] and res6["CER_list"] == res["CER_list"]
    # the NMS decoder path of the scripts (--NMS 0.5 --TH 0.3) runs too and writes its own CER file
    H.main(["--config", "tiny", "--weights", str(tmp_path / "checkpoint.pth"), "--images", str(img_dir), "--labels", str(tmp_path / "labels.json"),
            "--out", str(tmp_path / "stats"), "--dtype", "bf16", "--NMS", "0.5", "--TH", "0.3", "--size", "32", "--max_size", "256", "--batching", "padded"])
    assert (d / "cer_TH_0.3_NMS_0.5.txt").exists()


def test_data_parallel_two_ranks_on_one_gpu_equals_single_process():
    """SURVEY.md 8(e) correctness definition on hardware: two ranks of the REAL bench step (each with its own engine, shard of one
    global seeded batch) exchange their decode records, and rank 0 recomputes both shards itself: gathered == single-process
    records, bit for bit, in order.  Two ranks share the one GPU of the test box, so the process group is gloo (RCCL refuses
    duplicate devices); `--single-device` only points both ranks at cuda:0 -- their kernels interleave freely (round 2 needed a file
    lock here; DESIGN.md section 6 has the cause and the fix).  The sharding, the record exchange and the comparison are the code the
    8-GPU run takes."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "3",
           "--backend", "gloo", "--single-device", "--no-cpu-baseline", "--no-parity", "--min-seconds", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 6
    d = line["distributed"]
    assert (d["backend"], d["world_size"], d["dp_verified"]) == ("gloo", 2, True), d


def test_bench_gpus_n_starts_its_own_ranks():
    """The driver's command form, `python bench.py --gpus N` WITHOUT a launcher in the environment (round 3: it exited with "launch with
    torch.distributed.run"): bench.py re-executes itself through torch.distributed.run, rank 0 prints the one JSON line, dp_verified holds."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "3",
           "--backend", "gloo", "--single-device", "--no-cpu-baseline", "--no-parity", "--min-seconds", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    d = line["distributed"]
    assert line["n_gpus"] == 2 and (d["backend"], d["world_size"], d["dp_verified"]) == ("gloo", 2, True), d
    # a failing rank's exit code propagates
    bad = subprocess.run(cmd + ["--engine-opt", "no_such_attribute=1"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert bad.returncode != 0


@pytest.mark.parametrize("engine", [torch.float32, "f32s", torch.bfloat16], ids=["f32", "f32s", "bf16"])
def test_hip_graph_replay_reproduces_itself_after_an_eager_forward(engine):
    """Round 5: the forward captured in a HIP graph (torch.cuda.CUDAGraph) equals the eager forward bit for bit, and
    keeps doing so after eager forwards of the same engine ran in between.  Round 4 saw a replay diverge after an eager forward: the
    row-max vector of the two-stage head was initialised with hipMemsetD32Async, whose fill pattern is not part of the captured memset
    node's own state (a later replay filled zeros instead of -inf: scores of all-negative rows became 0).  It is a kernel now."""
    cfg = DTLRConfig.tiny()
    sd = weights.synthetic_state_dict(cfg, 0)
    from dtlr_amd.engine import DTLREngine
    dt = torch.float32 if engine == "f32s" else engine
    eng = DTLREngine(cfg, sd, "cuda:0", dt, split=engine == "f32s")
    lines = synth.noise_lines(3, 32, 256, seed=31)
    x0, x1 = torch.stack(lines[:2]).cuda(), torch.stack(lines[1:]).cuda()
    mask = torch.zeros((2, 32, 256), dtype=torch.bool, device="cuda:0")
    sx = x0.clone()

    def step():
        out = eng.forward(sx, mask, has_padding=False, return_debug=True)
        return {"scores": out["_debug"]["topk_scores"], "idx": out["_debug"]["topk_idx"], "logits": out["pred_logits"], "boxes": out["pred_boxes"]}

    def snap(r):
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in r.items()}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    e0 = snap(step())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        res = step()
    graph.replay()
    g1 = snap(res)
    eng.forward(x1, mask, has_padding=False)                       # eager forwards in between, another input
    eng.forward(sx, mask, has_padding=False)
    torch.cuda.synchronize()
    graph.replay()
    g2 = snap(res)
    sx.copy_(x1)
    graph.replay()
    g3 = snap(res)
    e3 = snap(step())
    for k in e0:
        assert torch.equal(e0[k], g1[k]), f"replay != eager at {k}"
        assert torch.equal(g1[k], g2[k]), f"replay after an eager forward != first replay at {k}"
        assert torch.equal(e3[k], g3[k]), f"replay on a new input != eager at {k}"
    assert not torch.equal(g1["logits"], g3["logits"])


def test_bench_accepts_a_user_checkpoint_and_an_image_folder(tmp_path):
    """BASELINE configs[2] the day its assets exist (round 5): `python bench.py --weights checkpoint.pth --images DIR` -- a reference-layout
    checkpoint ({"model": state_dict}; here the synthetic Latin weights with a 97-class head written to disk) and a folder of line images
    (PNG, sorted by name) through the reference's eval transform -- produces the one JSON line with the class count taken from the
    tensors, the observed backbone activation peak / MSDA choice, and the parity legs against the oracle ON THOSE ASSETS (fp32 engine:
    logits within 1e-3, identical strings on the same selection)."""
    import dataclasses
    import json
    import subprocess
    import sys
    from PIL import Image
    from tests.util import preproc_image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = dataclasses.replace(DTLRConfig.latin(), num_classes=97)
    torch.save({"model": weights.synthetic_state_dict(cfg, 2), "epoch": 7}, tmp_path / "checkpoint.pth")
    img_dir = tmp_path / "lines"
    img_dir.mkdir()
    for k in range(5):                                             # one more than the batch: the first 4 by name are taken
        Image.fromarray(preproc_image(96 + 8 * (k % 2), 1400 + 100 * k, 50 + k), "RGB").save(img_dir / f"line{k:02d}.png")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4", "--dtype", "f32", "--weights",
           str(tmp_path / "checkpoint.pth"), "--images", str(img_dir), "--no-cpu-baseline", "--no-other-dtypes", "--no-bs1", "--parity-lines", "2",
           "--min-seconds", "0", "--detail", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    stdout_lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(stdout_lines) == 1 and len(stdout_lines[0]) < 6000          # round 6: ONE compact line on stdout, the full result in --detail
    short = json.loads(stdout_lines[0])
    assert short["value"] > 0 and short["dtype"] == "f32" and short["by_dtype"]["f32"]["strings_teacher_forced"] == "2/2"
    line = json.load(open(tmp_path / "detail.json"))
    assert line["value"] == short["value"] and line["dtype"] == "f32" and line["config"]["global_batch"] == 4
    obs = line["observed_on_user_assets"]
    assert 0 < obs["backbone_activation_peak"] < 6e4 and len(obs["canvas"]) == 2 and 1333 <= obs["canvas"][1] <= 1344      # long side capped at 1333, canvas padded
    assert short["observed_on_user_assets"]["canvas"] == obs["canvas"]
    p = line["parity_vs_oracle"]
    assert "error" not in p, p
    assert p["parity_gate"] and p["teacher_forced"]["logit_err_max"] < LOGIT_TOL and p["teacher_forced"]["edit_distance"] == 0, p
    assert "free_running_v4" not in line                            # the v4 leg belongs to the synthetic Latin weights only


def test_ngram_emissions_and_rescoring_on_device(golden_dir):
    """SURVEY 8 f.4 on the GPU: the CTC-style emissions of the n-gram path (get_new_pred_logits, ngram/prediction_helpers.py:5-46) are
    produced by dtlr_blank_emissions (per-query sums chip-wide, the decoders' reading-order sort, one wave per row): (1) the vectors
    the reference's own function bodies produced (G8: sum, argmax row, the assembled strings through a fake decoder); (2) equal to
    the oracle for multiplier 1 and != 1 on both blank branches; (3) on real ENGINE outputs: emissions == oracle on the same
    logits, and get_ngram_prediction (with the self-contained lexicon beam decoder standing in for torchaudio's) returns the same
    string as the oracle's assembly driven by the same decoder."""
    import json
    from dtlr_amd import ngram as NG
    from oracle import dtlr_oracle as O
    from tests.util import fake_ctc_decoder, ngram_case
    g = json.load(open(os.path.join(golden_dir, "g8_ngram.json")))
    flags = ((True, False, True), (False, True, True), (True, True, False))
    for rec in g["cases"]:
        outputs, charset, ngc, ign = ngram_case(rec["seed"])
        dev = {k: v.cuda() for k, v in outputs.items()}
        new = NG.get_new_pred_logits(dev).cpu()
        assert abs(float(new.double().sum()) - rec["new_sum"]) < 1e-4 and new[0].argmax(-1).tolist() == rec["new_argmax"]
        assert (new - O.ngram_new_pred_logits(outputs)).abs().max() < 1e-6
        for k, (up, dg, ds) in enumerate(flags):
            assert NG.get_ngram_prediction(dev, fake_ctc_decoder(ngc), ign, charset, ngc, True, up, dg, ds) == rec[f"word_per_word_2_{k}"]
    # both blank branches, a multiplier, many lines
    r = np.random.Generator(np.random.PCG64(77))
    for C, bias, mult in ((23, -5.0, 1.0), (23, -1.0, 1.0), (166, -3.0, 1.7), (166, -7.0, 0.5)):
        out = {"pred_logits": torch.from_numpy((r.standard_normal((5, 900, C)) + bias).astype(np.float32)),
               "pred_boxes": torch.from_numpy(r.uniform(0.02, 0.98, (5, 900, 4)).astype(np.float32))}
        got = NG.get_new_pred_logits({k: v.cuda() for k, v in out.items()}, mult).cpu()
        want = O.ngram_new_pred_logits(out, mult)
        assert got.shape == want.shape and (got - want).abs().max() < 1e-6, (C, bias, mult)
    # engine outputs
    cfg = DTLRConfig.tiny(num_classes=23)
    sd = weights.synthetic_state_dict(cfg, 3)
    imgs = synth.stroke_lines(2, 32, [256, 224], seed=9)
    out = _model(cfg, sd)([i.cuda() for i in imgs])
    host = _cpu(out)
    assert (NG.get_new_pred_logits(out).cpu() - O.ngram_new_pred_logits(host)).abs().max() < 1e-6
    charset = [chr(ord("a") + i) for i in range(21)] + [" ", "-"]
    ngc = ["<ctc>"] + charset
    ign = [ngc.index(" ")]
    lex = {w: list(w) for w in ("ab", "abc", "cab", "bad", "dab", "a", "b")}
    dec = NG.LexiconCTCDecoder(ngc, lex, beam_size=8)
    for b in range(2):
        one_dev = {k: v[b:b + 1] for k, v in out.items() if k in ("pred_logits", "pred_boxes")}
        one_host = {k: v[b:b + 1] for k, v in host.items()}
        got = NG.get_ngram_prediction(one_dev, dec, ign, charset, ngc, True, True, False, True)
        want = O.ngram_word_per_word_pred_2(O.ngram_new_pred_logits(one_host), dec, ign, ngc, True, False, True)
        assert got == want


@pytest.mark.parametrize("half", ["bf16", "f16"])
def test_undamped_stress_weights_16bit_vs_fp32_engine(half):
    """Generator v3 (dtlr_amd/weights.py): decoder branches at natural gain + a prototype memory in the decoder FFNs -- the answer to
    "the x0.3 damping shrinks the error path".  Not a parity reference (its threshold units amplify even fp32 summation-order noise
    to ~1e-3), so the yardstick here is the exact-fp32 HIP engine on the same selection: per-query labels of the 16-bit engine
    identical on >= 99.5% of ALL queries (fp16: >= 99.9%), mean |logit error| bounded, everything finite."""
    from dtlr_amd.engine import DTLREngine
    from tests.util import query_decisions
    cfg = DTLRConfig.latin()
    sd = weights.synthetic_state_dict(cfg, 0, version=3)
    x = torch.stack(synth.stroke_lines(2, 128, 2048, seed=31) + synth.noise_lines(2, 128, 2048, seed=32)).cuda()
    mask = torch.zeros((4, 128, 2048), dtype=torch.bool, device="cuda")
    e32 = DTLREngine(cfg, sd, "cuda:0", torch.float32)
    ref = e32.forward(x, mask, return_debug=True, has_padding=False)
    e16 = DTLREngine(cfg, sd, "cuda:0", HALF[half])
    got = e16.forward(x, mask, forced_topk=ref["_debug"]["topk_idx"], has_padding=False)
    assert torch.isfinite(got["pred_logits"]).all() and torch.isfinite(got["pred_boxes"]).all()
    err = (got["pred_logits"].float() - ref["pred_logits"]).abs()
    rl, rm = query_decisions(ref["pred_logits"].cpu(), ref["pred_boxes"].cpu(), None)
    gl, _ = query_decisions(got["pred_logits"].float().cpu(), got["pred_boxes"].float().cpu(), None)
    agree = (rl == gl).float().mean().item()
    print(f"[v3 stress, {half}] logit err max {err.max().item():.3f} mean {err.mean().item():.5f}; labels agree on {agree:.5f} of all queries; "
          f"characters {int((rl >= 0).sum())}, designated-like margins median {rm[rl >= 0].median().item():.2f}")
    assert err.mean().item() < (0.03 if half == "bf16" else 0.004)            # measured 0.006 / 0.0008
    assert agree >= (0.995 if half == "bf16" else 0.999), agree


@pytest.mark.parametrize("half", ["bf16", "f16"])
def test_layer1_chain_backbone_equals_the_separate_convolutions(half):
    """engine.use_l1_chain (layer1's 1x1 convolutions chained: dtlr_gemm_kres_chain) against the same engine with one launch per
    convolution, on the backbone maps C3 / C4 / C5 of stroke + noise lines.  The chain changes ONE rounding -- the first bottleneck's
    shortcut convolution is no longer rounded to 16 bits before the add -- so the maps agree to a few 16-bit ulps of their scale, not
    bit for bit; both sit equally close to the fp32 engine's maps."""
    from dtlr_amd.engine import DTLREngine
    cfg = DTLRConfig.latin()
    sd = weights.synthetic_state_dict(cfg, 0)
    x = torch.stack(synth.stroke_lines(2, 128, 2048, seed=41) + synth.noise_lines(1, 128, 2048, seed=42)).cuda()
    eng = DTLREngine(cfg, sd, "cuda:0", HALF[half])
    assert eng.use_l1_chain and eng.use_l1_chain_out
    got = [f.float() for f in eng.backbone(x)]
    eng.use_l1_chain = False
    sep = [f.float() for f in eng.backbone(x)]
    ref = DTLREngine(cfg, sd, "cuda:0", torch.float32).backbone(x)
    u = 2.0 ** (-8 if half == "bf16" else -11)
    for g, s_, r in zip(got, sep, ref):
        scale = r.abs().max().item()
        eg, es = (g - r).abs(), (s_ - r).abs()
        print(f"[l1 chain, {half}] map {tuple(g.shape)}: |chain - sep| max {(g - s_).abs().max().item():.4g}, vs fp32: chain {eg.max().item():.4g} / "
              f"{eg.mean().item():.4g}, separate {es.max().item():.4g} / {es.mean().item():.4g} (scale {scale:.3g})")
        assert torch.isfinite(g).all()
        assert eg.mean().item() <= 1.1 * es.mean().item() + 1e-6                   # no further from the exact maps than the separate launches
        assert eg.max().item() <= 64 * u * scale
    # the chain without its last link (layer2.0.conv1 from its own launch): bit-identical maps (that link changes no rounding)
    eng.use_l1_chain, eng.use_l1_chain_out = True, False
    for g, h in zip(got, eng.backbone(x)):
        assert torch.equal(g, h.float())


@pytest.mark.parametrize("engine", ["bf16", "f32s", "f32"])
def test_side_stream_schedule_is_bit_identical_to_the_one_stream_schedule(engine):
    """Round 6: DTLREngine.overlap_streams (off by default: measured 0.7 % slower at B = 32) runs input_proj + GroupNorm of levels 0 / 1 under
    layer3 / layer4, the shortcut convolutions under conv1 -> conv2 and the decoder's six value projections under the two-stage selection,
    on side HIP streams (fork / join with stream waits, buffers allocated before the fork).
    Same kernels, same arguments: every output is bit-identical to the one-stream schedule -- tiny model with mixed widths (padded),
    full-size lines, repeated forwards (a missing join would show up as a difference in some repetition), and a caller-side stream."""
    from dtlr_amd.engine import DTLREngine
    dt = HALF[engine] if engine in HALF else torch.float32
    for cfg, H, W, n in ((DTLRConfig.tiny(), 32, 256, 3), (DTLRConfig.latin(), 128, 2048, 2)):
        sd = weights.synthetic_state_dict(cfg, 0)
        eng = DTLREngine(cfg, sd, "cuda:0", dt, split=engine == "f32s")
        assert not eng.overlap_streams                              # measured slower than the one-stream schedule at B = 32: off by default
        x = torch.stack(synth.stroke_lines(n - 1, H, W, seed=61) + synth.noise_lines(1, H, W, seed=62)).cuda()
        mask = torch.zeros((n, H, W), dtype=torch.bool, device="cuda:0")
        for padded in (False, True):
            if padded:
                mask[0, :, W - W // 4:] = True
                x[0, :, :, W - W // 4:] = 0
            eng.overlap_streams = False
            want = eng.forward(x, mask, has_padding=padded, return_debug=True)
            torch.cuda.synchronize()
            eng.overlap_streams = True
            for rep in range(6 if cfg.enc_layers == 2 else 2):
                got = eng.forward(x, mask, has_padding=padded, return_debug=True)
                for k in ("pred_logits", "pred_boxes"):
                    assert torch.equal(got[k], want[k]), (engine, padded, rep, k)
                for k in ("src", "memory", "topk_idx"):
                    assert torch.equal(got["_debug"][k], want["_debug"][k]), (engine, padded, rep, k)
            caller = torch.cuda.Stream()
            caller.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(caller):
                got = eng.forward(x, mask, has_padding=padded)
            caller.synchronize()
            assert torch.equal(got["pred_logits"], want["pred_logits"])
        del eng
        torch.cuda.empty_cache()


def test_split_engine_multi_slice_projections_equal_the_separate_launches():
    """Round 6: the split engine's two uses of dtlr_gemm_k256s_multi inside the model, on two bench-sized lines (the encoder form needs
    S % 64 == 0: 5440 at 128x2048).  (i) decoder value_proj(memory) as six slices of one launch (default ON) against the tiled split GEMM:
    the same products in another summation order -- logits within 2e-5; (ii) the encoder form (value_proj + [offsets | logits] of src as
    three slices, default OFF: measured slower than its parts) against the default path: encoder memory and logits within 2e-5."""
    from dtlr_amd.engine import DTLREngine
    cfg = DTLRConfig.latin()
    sd = weights.synthetic_state_dict(cfg, 0)
    x = torch.stack(synth.stroke_lines(1, 128, 2048, seed=51) + synth.noise_lines(1, 128, 2048, seed=52)).cuda()
    mask = torch.zeros((2, 128, 2048), dtype=torch.bool, device="cuda:0")
    eng = DTLREngine(cfg, sd, "cuda:0", torch.float32, split=True)
    assert eng.use_k256s_multi and not eng.use_k256s_multi_enc
    base = eng.forward(x, mask, has_padding=False, return_debug=True)
    idx = base["_debug"]["topk_idx"]
    eng.use_k256s_multi = False
    tiled = eng.forward(x, mask, has_padding=False, forced_topk=idx, return_debug=True)
    eng.use_k256s_multi, eng.use_k256s_multi_enc = True, True
    enc3 = eng.forward(x, mask, has_padding=False, forced_topk=idx, return_debug=True)
    assert "enc0.attn.ow.k256sm" in eng.w, "the encoder form did not run"
    for name, other in (("decoder slices vs tiled GEMM", tiled), ("encoder slices vs separate launches", enc3)):
        dm = (other["_debug"]["memory"] - base["_debug"]["memory"]).abs().max().item()
        dl = (other["pred_logits"] - base["pred_logits"]).abs().max().item()
        db = (other["pred_boxes"] - base["pred_boxes"]).abs().max().item()
        print(f"[k256s_multi in the model] {name}: memory {dm:.2e}, logits {dl:.2e}, boxes {db:.2e}")
        assert dm < 2e-5 and dl < 2e-5 and db < 2e-6, (name, dm, dl, db)
    # padded batch: the six decoder slices carry the row mask
    mask[1, :, 1500:] = True
    x[1, :, :, 1500:] = 0
    p_on = eng.forward(x, mask, has_padding=True, return_debug=True)
    eng.use_k256s_multi = False
    p_off = eng.forward(x, mask, has_padding=True, forced_topk=p_on["_debug"]["topk_idx"])
    assert (p_on["pred_logits"] - p_off["pred_logits"]).abs().max().item() < 2e-5


@pytest.mark.parametrize("half", ["bf16", "f16"])
def test_layer2_cat_backbone_close_to_the_separate_convolutions(half):
    """engine.use_l2_cat (layer2.0's strided shortcut convolution as extra K columns of its tail GEMM: dtlr_gemm_kres_cat_s2) against the
    same engine with the shortcut as its own launch: the only numerical change is that the shortcut is no longer rounded to 16 bits
    before the add, so the maps agree to a few 16-bit ulps and are no further from the fp32 engine's; full canvas and the odd-sized
    eval canvas (83 x 1328 -> a 21 x 332 layer1 map)."""
    from dtlr_amd.engine import DTLREngine
    cfg = DTLRConfig.latin()
    sd = weights.synthetic_state_dict(cfg, 0)
    eng = DTLREngine(cfg, sd, "cuda:0", HALF[half])
    e32 = DTLREngine(cfg, sd, "cuda:0", torch.float32)
    u = 2.0 ** (-8 if half == "bf16" else -11)
    for (H, W) in ((128, 2048), (83, 1328)):
        x = torch.stack(synth.stroke_lines(2, H, W, seed=51) + synth.noise_lines(1, H, W, seed=52)).cuda()
        eng.use_l2_cat = True
        got = [f.float() for f in eng.backbone(x)]
        eng.use_l2_cat = False
        sep = [f.float() for f in eng.backbone(x)]
        ref = e32.backbone(x)
        for g, s_, r in zip(got, sep, ref):
            scale = r.abs().max().item()
            eg, es = (g - r).abs(), (s_ - r).abs()
            print(f"[l2 cat, {half}, {H}x{W}] map {tuple(g.shape)}: |cat - sep| max {(g - s_).abs().max().item():.4g}, vs fp32: cat {eg.max().item():.4g} / "
                  f"{eg.mean().item():.4g}, separate {es.max().item():.4g} / {es.mean().item():.4g} (scale {scale:.3g})")
            assert g.shape == s_.shape and torch.isfinite(g).all()
            assert eg.mean().item() <= 1.1 * es.mean().item() + 1e-6
            assert eg.max().item() <= 64 * u * scale


def test_rccl_collectives_on_a_one_rank_group():
    """The `nccl` (= RCCL) branch of dtlr_amd.dist -- all_gather_into_tensor of the decode records, the MAX all-reduce of the timing,
    the barrier -- on a world-size-1 group on the one GPU of the test box: the code path of the 8-GPU job has then executed before
    it is ever launched on 8 GPUs (a subprocess: the process group must not leak into the other tests)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, torch\n"
        "from dtlr_amd import dist as D\n"
        "import torch.distributed as td\n"
        "r, l, w = D.init_from_env('nccl', force_group=True)\n"
        "assert td.is_initialized() and td.get_backend() == 'nccl' and (r, l, w) == (0, 0, 1)\n"
        "lab = torch.arange(5 * 900, dtype=torch.int32, device='cuda:0').view(5, 900)\n"
        "ln = torch.tensor([3, 0, 900, 7, 1], dtype=torch.int32, device='cuda:0')\n"
        "a, b = D.all_gather_records(lab, ln, 5, force_collective=True)\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(a, lab) and torch.equal(b, ln)\n"
        "assert D.max_over_ranks(1.25, torch.device('cuda:0'), force_collective=True) == 1.25\n"
        "td.barrier(); D.finalize(); print('RCCL_OK')\n")
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29731",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


@pytest.mark.parametrize("kind,conf,n", [("bf16", "latin", 50), ("f16", "latin", 50), ("f32s", "latin", 16), ("f32", "latin", 8),
                                         ("bf16-gather", "latin", 30), ("bf16", "chinese", 20), ("bf16", "swin", 20)])
def test_two_processes_on_one_gpu_are_bit_reproducible(kind, conf, n):
    """Round 2's open issue: with two processes driving ONE GPU, ~10% of forwards differed from the same process's own earlier result
    (lanes 48..63 of scattered waves of the decoder's deformable-sampling kernel dropped corner terms).  Two
    UNLOCKED workers, n forwards each of the same batch, no synchronisation inside a forward: every forward of both processes must be
    bit-identical.  Round 4 extends the round-3 pair (bf16 / f16 Latin) to every engine and kernel family that could hold a long-lived
    lane mask: the exact-fp32 and split-fp32 engines, the encoder forced onto the gather kernel, the 7356-class head, the Swin backbone."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tests", "contention_worker.py"), str(n), kind, conf]
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root) for _ in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
    assert all(o["forwards"] == n and o["distinct"] == 1 for o in outs), outs
    assert outs[0]["digest"] == outs[1]["digest"], outs


# ---- Swin backbones ---------------------------------------------------------------------------
def test_swin_backbone_fp32_vs_reference_golden_and_oracle(golden_dir):
    """engine.backbone_swin (patch embed, window attention with shift / padding / relative bias on the exact-fp32 matrix cores, MLP
    with the GELU epilogue, patch merging) == the reference's SwinTransformer outputs (G7a: odd sizes) and the oracle on a second
    shape whose stages are all multiples of the window (no padding path) and one with a single window per axis."""
    from dtlr_amd.engine import DTLREngine
    from oracle import dtlr_oracle as O
    from tests.golden.make_golden_swin import custom_cfg
    g = np.load(os.path.join(golden_dir, "g7_swin.npz"))
    cfg = custom_cfg()
    sd = weights.synthetic_state_dict(cfg, 0)
    eng = DTLREngine(cfg, sd, "cuda:0", torch.float32)
    x = torch.stack(synth.noise_lines(2, 37, 90, seed=71))
    feats = eng.backbone_swin(x.cuda())
    for i, f in enumerate(feats):
        want = torch.from_numpy(g[f"a_feat{i}"]).permute(0, 2, 3, 1)
        assert f.shape == want.shape
        assert (f.cpu() - want).abs().max() < 2e-4, i
    for (h, w) in ((64, 128), (16, 16), (33, 260)):
        x = torch.stack(synth.noise_lines(2, h, w, seed=h))
        want = O.swin_body(x, sd, cfg.swin_params())
        got = eng.backbone_swin(x.cuda())
        for a, b in zip(got, want):
            assert (a.cpu() - b.permute(0, 2, 3, 1)).abs().max() < 2e-4, (h, w)


def test_swin_t_full_model_fp32_vs_reference_golden(golden_dir):
    """DINO with backbone = 'swin_T_224_1k' (backbone.py:172-205) against the reference's build_dino outputs (G7b), selection pinned."""
    from tests.golden.make_golden_swin import swin_t_cfg
    g = np.load(os.path.join(golden_dir, "g7_swin.npz"))
    cfg = swin_t_cfg()
    sd = weights.synthetic_state_dict(cfg, 0)
    m = _model(cfg, sd)
    imgs = synth.stroke_lines(1, 64, 256, seed=5) + synth.noise_lines(1, 48, 200, seed=6)
    out = m([i.cuda() for i in imgs], forced_topk=torch.from_numpy(g["b_topk_idx"].astype(np.int64)).cuda(), return_debug=True)
    assert (out["_debug"]["topk_scores"].cpu() - torch.from_numpy(g["b_topk_scores"])).abs().max() < 5e-4
    assert (out["_debug"]["memory"][:, ::7].cpu() - torch.from_numpy(g["b_memory"])).abs().max() < 5e-4
    assert (out["pred_logits"].cpu() - torch.from_numpy(g["b_pred_logits"])).abs().max() < LOGIT_TOL
    assert (out["pred_boxes"].cpu() - torch.from_numpy(g["b_pred_boxes"])).abs().max() < BOX_TOL


@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("embed,heads", [(64, (2, 4, 8, 16)), (96, (3, 6, 12, 24))])
def test_swin_backbone_bf16_close_to_oracle(embed, heads, half):
    """16-bit engines (MFMA window attention, MFMA GEMMs) on a Swin backbone: features close to the fp32 oracle (relative error of a
    few ulps per block, 8 blocks).  embed 96 is swin_T's width (swin_transformer.py:686-692): its K = 96 projections run on
    zero-padded K = 128 operands (DTLREngine._pack_swin), which round 2 refused in 16 bits."""
    import dataclasses
    from dtlr_amd.engine import DTLREngine
    from oracle import dtlr_oracle as O
    cfg = dataclasses.replace(DTLRConfig.tiny(), backbone="swin_custom", swin_embed_dim=embed, swin_depths=(2, 2, 2, 2),
                              swin_num_heads=heads, swin_window=7)
    sd = weights.synthetic_state_dict(cfg, 2)
    eng = DTLREngine(cfg, sd, "cuda:0", HALF[half])
    x = torch.stack(synth.noise_lines(2, 60, 250, seed=3))
    want = O.swin_body(x, sd, cfg.swin_params())
    got = eng.backbone_swin(x.cuda())
    for a, b in zip(got, want):
        b = b.permute(0, 2, 3, 1)
        rel = (a.float().cpu() - b).abs().mean() / b.abs().mean()
        assert rel < (0.03 if half == "bf16" else 0.005), rel.item()


@pytest.mark.parametrize("half", ["bf16", "f16"])
def test_swin_t_full_model_16bit_runs_and_tracks_fp32(half):
    """The reference's first Swin variant, swin_T_224_1k (backbone.py:172-205), end to end on the 16-bit engines against the exact-fp32
    HIP engine (itself pinned to the reference's build_dino outputs, G7b) with the same selection."""
    import dataclasses
    from dtlr_amd.engine import DTLREngine
    cfg = dataclasses.replace(DTLRConfig.tiny(), backbone="swin_T_224_1k")
    sd = weights.synthetic_state_dict(cfg, 0)
    x = torch.stack(synth.stroke_lines(2, 64, 256, seed=5)).cuda()
    mask = torch.zeros((2, 64, 256), dtype=torch.bool, device="cuda")
    ref = DTLREngine(cfg, sd, "cuda:0", torch.float32).forward(x, mask, return_debug=True)
    got = DTLREngine(cfg, sd, "cuda:0", HALF[half]).forward(x, mask, forced_topk=ref["_debug"]["topk_idx"])
    assert torch.isfinite(got["pred_logits"]).all()
    err = (got["pred_logits"].float() - ref["pred_logits"]).abs()
    print(f"[swin_T {half}] logit err max {err.max().item():.4f} mean {err.mean().item():.5f}")
    assert err.max().item() < (0.5 if half == "bf16" else 0.08) and err.mean().item() < (0.05 if half == "bf16" else 0.008)


def test_msda_kernel_choice_follows_the_far_sample_probe():
    """An encoder layer whose sampling offsets leave the LDS kernel's staged windows is switched to the gather kernel by the probe
    (DTLREngine._msda_mode / _calibrate_msda), one with small offsets stays on the LDS kernel; both match the oracle either way."""
    from oracle import dtlr_oracle as O
    cfg = DTLRConfig.tiny()
    sd = weights.synthetic_state_dict(cfg, 3)
    far_layer = "transformer.encoder.layers.1.self_attn.sampling_offsets.bias"
    sd = dict(sd)
    b = torch.zeros_like(sd[far_layer])
    b[0::2] = torch.linspace(-60.0, 60.0, b.numel() // 2)          # x offsets of tens of pixels (y stays inside the 4-row maps)
    sd[far_layer] = b
    imgs = synth.noise_lines(2, 32, 2048, seed=9)       # level 0 is 256 columns wide: several window tiles
    m = _model(cfg, sd)
    out = m([i.cuda() for i in imgs], return_debug=True)
    st = {k[0]: v for k, v in m.engine()._msda_state.items()}
    assert st["enc1.attn"]["mode"] == "gather" and min(st["enc1.attn"]["far"].values()) > 0.05, st
    assert st["enc0.attn"]["mode"] == "lds" and st["enc0.attn"]["halo"] == 8 and st["enc0.attn"]["far"][8] < 0.012, st
    # the choice is a function of (weights, canvas shape): a second engine with the same weights, fed DIFFERENT data first, agrees
    m2 = _model(cfg, sd)
    m2([i.cuda() for i in synth.stroke_lines(2, 32, 2048, seed=77)])
    assert {k: v["mode"] for k, v in m2.engine()._msda_state.items()} == {k: v["mode"] for k, v in m.engine()._msda_state.items()}
    ref = O.dino_forward(sd, cfg, imgs, forced_topk=out["_debug"]["topk_idx"].cpu())
    assert (out["pred_logits"].cpu() - ref["pred_logits"]).abs().max() < LOGIT_TOL
    assert (out["pred_boxes"].cpu() - ref["pred_boxes"]).abs().max() < BOX_TOL
