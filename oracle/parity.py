"""Engine-vs-oracle parity legs on a batch of lines (TEST INFRASTRUCTURE, like everything under oracle/: imported by tests/,
bench.py's checker legs and __graft_entry__.smoke() only -- never by the product path).

Two comparisons, both against the CPU oracle's forward of the same lines (oracle/dtlr_oracle.py, pinned to the reference through
tests/golden):

  free-running     the engine with ITS OWN two-stage selection against the oracle with ITS OWN selection: decoded strings compared as
                   they come out (what a user of evaluation.py sees).  No tolerance, no accounting.
  teacher-forced   the oracle's selection + decoder re-run on the engine's selection (`dino_forward(resume=..., forced_topk=...)`):
                   isolates arithmetic error (max |logit| / |box| difference) from the discrete effect below.

Why both: tgt_embed row i belongs to the query of selection RANK i (embed_init_tgt, deformable_transformer.py:354-355), so two
tokens whose two-stage scores differ by less than the score error of an implementation trade ranks, i.e. trade content queries.
That happens between any two fp32 implementations (the reference on CPU vs on CUDA too).  `self_sensitivity` puts a number on it
for the weights at hand: the ORACLE against ITSELF with its selection scores perturbed by the engine's measured score error.

Weights: the teacher-forced budgets are stated for generator v2 (the goldens' weights), whose characters are planted per selection RANK --
which makes v2's free-running strings a measure of rank swaps, not of arithmetic (a 16-bit engine: ~90% CER; the oracle against itself at
1e-5 score noise: half of the strings).  The free-running leg that says something about the engine runs on generator v4
(dtlr_amd/weights.py: identical content queries, characters read from the image by one-shot detector units -- rank-invariant like a trained
recogniser): there the strings must simply be equal for an fp32-grade engine.

The a-priori error budgets (north_star: logits within 1e-3 for the fp32-grade engines; the stated bounds of the 16-bit engines) are
the gate -- fixed numbers, not derived from the measured error."""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch

from . import dtlr_oracle as O
from .compare import query_decisions

# a-priori budgets per engine: max |logit error| / max |box error| against the oracle on the same selection.  fp32-grade engines: the
# north_star tolerance; 16-bit engines: the bounds tests/test_gpu_model.py asserts (BF16_LOGIT_BOUND / F16_LOGIT_BOUND, Latin).
LOGIT_BUDGET = {"f32": 1e-3, "f32s": 1e-3, "f16": 0.06, "bf16": 0.3}
BOX_BUDGET = {"f32": 1e-4, "f32s": 1e-4, "f16": 4e-3, "bf16": 2e-2}
CHINESE_LOGIT_BUDGET = {"f32": 1e-3, "f32s": 1e-3, "f16": 0.08, "bf16": 0.4}


def _sci(v: float) -> float:
    return float(f"{v:.3e}")


def _edit(a: Sequence[Sequence[int]], b: Sequence[Sequence[int]]):
    dist = sum(O.levenshtein(x, y) for x, y in zip(a, b))
    return dist, sum(len(x) for x in a), sum(1 for x, y in zip(a, b) if list(x) == list(y))


class OracleBatch:
    """The oracle's free-running forward of `x` [n,3,H,W] (+ mask) once; every engine is then compared against it."""

    def __init__(self, cfg, sd, x: torch.Tensor, mask: torch.Tensor, threads: Optional[int] = None, eps: Optional[float] = None):
        self.cfg, self.sd, self.eps = cfg, sd, eps
        torch.set_num_threads(threads or min(16, os.cpu_count() or 8))
        self.free = O.dino_forward(sd, cfg, x.float().cpu(), mask=mask.cpu(), return_debug=True)
        self.debug = self.free["_debug"]
        self.strings = O.decode_blank(self.free, eps)
        self.n = int(x.shape[0])

    def teacher_forced(self, topk_idx: torch.Tensor) -> Dict[str, torch.Tensor]:
        """the oracle's selection-dependent part (gather, decoder, heads) on `topk_idx` [n, nq]; encoder output reused"""
        return O.dino_forward(self.sd, self.cfg, None, forced_topk=topk_idx.cpu().long(), resume=self.debug)

    def self_sensitivity(self, score_err: float, seed: int = 0) -> Dict[str, object]:
        """the oracle against itself with its two-stage scores perturbed by uniform +-score_err noise: how many of ITS strings survive
        the rank swaps a second correct implementation with that score error would make"""
        sc = self.debug["topk_scores"]
        g = torch.Generator().manual_seed(seed)
        noise = (torch.rand(sc.shape, generator=g) * 2 - 1) * score_err
        idx = torch.topk(sc + noise, self.cfg.num_queries, dim=1)[1]
        alt = O.decode_blank(self.teacher_forced(idx), self.eps)
        dist, n, same = _edit(self.strings, alt)
        return {"score_noise": _sci(score_err), "strings_identical": f"{same}/{self.n}", "cer": round(dist / max(n, 1), 5),
                "rank_slots_changed": int((idx != self.debug["topk_idx"]).sum()),
                "lines_with_identical_selection": int((idx == self.debug["topk_idx"]).all(1).sum())}

    def compare(self, engine: str, logits: torch.Tensor, boxes: torch.Tensor, topk_idx: torch.Tensor,
                topk_scores: Optional[torch.Tensor] = None, chinese: bool = False, sensitivity: bool = True,
                budgeted: bool = True) -> Dict[str, object]:
        """`logits` / `boxes` / `topk_idx` (/ `topk_scores`) = the engine's FREE-RUNNING outputs for the same n lines.  `budgeted` False:
        no error budget is applied (generator v4's detector units amplify a hidden-state error ~1e3-fold by design -- its purpose is the
        free-running string comparison; the budgets belong to the v2 weights of the goldens)."""
        got = {"pred_logits": logits.float().cpu(), "pred_boxes": boxes.float().cpu()}
        idx = topk_idx.cpu().long()
        got_strings = O.decode_blank(got, self.eps)
        # ---- free-running: strings as they are
        dist, nchar, same = _edit(self.strings, got_strings)
        sel_same = (idx == self.debug["topk_idx"]).all(1)
        same_given_sel = sum(1 for b in range(self.n) if bool(sel_same[b]) and list(self.strings[b]) == list(got_strings[b]))
        free = {"strings_identical_free_running": f"{same}/{self.n}", "cer_free_running": round(dist / max(nchar, 1), 5),
                "edit_distance": dist, "chars_oracle": nchar,
                "lines_with_identical_selection": int(sel_same.sum()),
                "strings_identical_given_identical_selection": f"{same_given_sel}/{int(sel_same.sum())}",
                "rank_slots_changed": int((idx != self.debug["topk_idx"]).sum())}
        score_err = None
        if topk_scores is not None:
            score_err = (topk_scores.float().cpu() - self.debug["topk_scores"]).abs().max().item()
            free["two_stage_score_err_max"] = _sci(score_err)
            if sensitivity:
                free["oracle_vs_itself_at_that_score_error"] = self.self_sensitivity(score_err)
        # ---- teacher-forced: arithmetic error on the engine's own selection
        ref = self.teacher_forced(idx)
        E = (got["pred_logits"] - ref["pred_logits"]).abs().max().item()
        Eb = (got["pred_boxes"] - ref["pred_boxes"]).abs().max().item()
        Ecx = (got["pred_boxes"][..., 0] - ref["pred_boxes"][..., 0]).abs().max().item()
        ref_strings = O.decode_blank(ref, self.eps)
        tdist, tn, tsame = _edit(ref_strings, got_strings)
        rl, rm = query_decisions(ref["pred_logits"], ref["pred_boxes"], self.eps)
        gl, _ = query_decisions(got["pred_logits"], got["pred_boxes"], self.eps)
        lb = (CHINESE_LOGIT_BUDGET if chinese else LOGIT_BUDGET).get(engine)
        bb = BOX_BUDGET.get(engine)
        tf = {"logit_err_max": _sci(E), "box_err_max": _sci(Eb), "cx_err_max": _sci(Ecx),
              "logit_err_mean": _sci((got["pred_logits"] - ref["pred_logits"]).abs().mean().item()),
              "logit_budget": lb, "box_budget": bb, "within_budget": bool(lb is not None and E <= lb and Eb <= bb),
              "strings_identical_same_selection": f"{tsame}/{self.n}", "cer_same_selection": round(tdist / max(tn, 1), 5),
              "edit_distance": tdist, "chars_oracle": tn, "label_flips": int((rl != gl).sum()),
              "min_oracle_margin_on_flipped": (_sci(rm[rl != gl].min().item()) if bool((rl != gl).any()) else None)}
        if not budgeted:
            for k in ("logit_budget", "box_budget", "within_budget"):
                tf.pop(k)
            return {"lines": self.n, "free_running": free, "teacher_forced": tf}
        return {"lines": self.n, "free_running": free, "teacher_forced": tf,
                # the north_star statement for this engine: logits within its budget AND identical strings on the same selection
                "parity_gate": bool(tf["within_budget"] and tdist == 0) if engine in ("f32", "f32s") else bool(tf["within_budget"])}


def free_running_strings_equal(ob: OracleBatch, got_strings: List[List[int]], topk_idx: torch.Tensor) -> List[int]:
    """rows whose selection equals the oracle's but whose strings differ (must be empty for an fp32-grade engine)"""
    sel_same = (topk_idx.cpu().long() == ob.debug["topk_idx"]).all(1)
    return [b for b in range(ob.n) if bool(sel_same[b]) and list(ob.strings[b]) != list(got_strings[b])]
