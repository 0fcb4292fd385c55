"""Shared helpers for the tests (test infrastructure; may import the oracle)."""
import ctypes

import numpy as np
import torch


def msda_inputs(N, M, D, Lq, P, shapes, seed, lo=0.0, hi=1.0, value_scale=1.0, dtype=np.float32):
    """Same generator as tests/golden/make_golden.py::msda_inputs (numpy PCG64)."""
    r = np.random.Generator(np.random.PCG64(seed))
    S = sum(h * w for h, w in shapes)
    L = len(shapes)
    value = (r.random((N, S, M, D), dtype=np.float32) * 2 - 1) * value_scale
    loc = r.uniform(lo, hi, (N, Lq, M, L, P, 2)).astype(np.float32)
    aw = r.random((N, Lq, M, L, P), dtype=np.float32) + 1e-5
    aw = (aw / aw.sum((-1, -2), keepdims=True)).astype(np.float32)
    shp = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    return (torch.from_numpy(value.astype(dtype)), shp, lsi, torch.from_numpy(loc.astype(dtype)), torch.from_numpy(aw.astype(dtype)))


def c_oracle_msda(clib, value, shapes, lsi, loc, attn):
    """Run oracle/msda_ref.c on CPU tensors."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = torch.empty((N, Lq, M * D), dtype=value.dtype)
    fn = clib.msda_ref_forward_f64 if value.dtype == torch.float64 else clib.msda_ref_forward_f32
    fn.restype = ctypes.c_int
    vp = ctypes.c_void_p
    rc = fn(vp(value.contiguous().data_ptr()), vp(shapes.contiguous().data_ptr()), vp(lsi.contiguous().data_ptr()),
            vp(loc.contiguous().data_ptr()), vp(attn.contiguous().data_ptr()),
            N, S, M, D, L, Lq, P, vp(out.data_ptr()))
    assert rc == 0
    return out


def selection_is_valid(idx, scores, k, tol):
    """Tie-aware check of a top-k selection `idx` [B,k] against fp32 `scores` [B,S]: every selected
    score >= (k-th largest - tol), every unselected <= (k-th largest + tol), order descending within tol."""
    for b in range(scores.shape[0]):
        s = scores[b]
        kth = torch.topk(s, k)[0][-1]
        sel = s[idx[b].long()]
        if not bool((sel >= kth - tol).all()):
            return False
        m = torch.ones_like(s, dtype=torch.bool)
        m[idx[b].long()] = False
        if m.any() and not bool((s[m] <= kth + tol).all()):
            return False
        if not bool((sel[:-1] >= sel[1:] - tol).all()):
            return False
        if len(set(idx[b].tolist())) != k:
            return False
    return True


def preproc_image(h, w, seed):
    """Seeded uint8 RGB test image [h, w, 3] shared by tests/golden/make_golden_preproc.py and the preprocessing tests:
    uniform noise, or (odd seeds) stroke-like light paper with dark runs."""
    g = np.random.Generator(np.random.PCG64(seed))
    base = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if seed % 2:
        base = np.where(g.random((h, w, 1)) < 0.15, base // 4, 200 + base // 5).astype(np.uint8)
    return base


def ctc_case(seed, B, nq, C, bias, lmax):
    """Seeded head outputs + label sequences for the CTC-loss tests (shared with tests/golden/make_golden_ctc.py):
    logits ~ N(bias, 1) with a few confident queries per line, boxes uniform, ragged label sequences incl. an empty one and
    repeated characters."""
    g = np.random.Generator(np.random.PCG64(1000 + seed))
    logits = (g.standard_normal((B, nq, C)) + bias).astype(np.float32)
    for b in range(B):
        hot = g.choice(nq, size=max(2, nq // 6), replace=False)
        logits[b, hot, g.integers(0, C, hot.shape[0])] += 9.0
    boxes = g.uniform(0.02, 0.98, (B, nq, 4)).astype(np.float32)
    labels = []
    for b in range(B):
        L = int(g.integers(1, min(lmax, nq) + 1))
        seq = g.integers(0, C, L).tolist()
        if L > 3:
            seq[2] = seq[1]                                        # a repeated character: needs the blank between them
        labels.append(seq)
    if B > 2:
        labels[-1] = []                                            # an empty transcription
    return {"pred_logits": torch.from_numpy(logits), "pred_boxes": torch.from_numpy(boxes)}, labels


# ---- margin-aware comparison of decoded outputs -------------------------------------------------------------------------------
def query_decisions(logits, boxes, eps):
    """Per-query view of the blank/argmax decoder (evaluation.py:116-158 / dino.py:466-502), BEFORE the reading-order sort:
    label [B,nq] (class index, -1 = blank) and a decision margin [B,nq] in LOGIT units: how far the per-query logits may move
    (max-abs, all classes at once) before the decision can change -- min(top1 - top2 logit gap, |logit(top prob) - logit(blank prob)|),
    the second term halved because a uniform shift moves the blank channel (1 - sum p) against the class channel."""
    logits = logits.float()
    C = logits.shape[-1]
    p = torch.sigmoid(logits)
    s = p.sum(-1)
    e = eps if eps is not None else 0.03 / C
    small = s < 1 - e
    ptop, arg = p.max(-1)
    blank = torch.where(small, 1 - s, torch.full_like(s, e))
    top = torch.where(small, ptop, (1 - e) * ptop / s)
    label = torch.where(blank >= top, torch.full_like(arg, -1), arg)
    t2 = logits.topk(2, -1)[0]
    lg = lambda x: torch.log(x.clamp(1e-12, 1 - 1e-7) / (1 - x.clamp(1e-12, 1 - 1e-7)))
    m_blank = (lg(top) - lg(blank)).abs() * 0.5
    margin = torch.where(label >= 0, torch.minimum(t2[..., 0] - t2[..., 1], m_blank), m_blank)
    return label, margin


def safe_reading(labels, margins, cx, logit_bound, cx_bound):
    """Reading-order strings restricted to SAFE queries: decision margin > 2 * logit_bound, and (for the order) no other non-blank
    query closer than 2 * cx_bound in cx.  Returns (list of index tensors = safe non-blank queries in cx order, safe mask [B,nq])."""
    out, safe_all = [], labels.new_zeros(labels.shape, dtype=torch.bool)
    for b in range(labels.shape[0]):
        safe = margins[b] > 2 * logit_bound
        safe_all[b] = safe
        cand = torch.nonzero((labels[b] >= 0) | ~safe).flatten()          # every query that may print a character
        order = cand[torch.argsort(cx[b, cand], stable=True)]
        c = cx[b, order]
        close = torch.zeros_like(c, dtype=torch.bool)
        if len(c) > 1:
            gap = (c[1:] - c[:-1]) < 2 * cx_bound
            close[1:] |= gap
            close[:-1] |= gap
        keep = ~close & safe[order] & (labels[b, order] >= 0)
        out.append(order[keep])
    return out, safe_all


def compare_decoded(ref_logits, ref_boxes, got_logits, got_boxes, eps, logit_bound, cx_bound):
    """Margin-aware equality of two decodes of the same queries: on every query whose reference margin exceeds 2 * logit_bound the
    labels must be identical, and the reading-order strings restricted to the safe, cx-separated non-blank queries must be identical
    (CER == 0 on them).  Returns statistics for the caller to assert on / print."""
    rl, rm = query_decisions(ref_logits, ref_boxes, eps)
    gl, _ = query_decisions(got_logits, got_boxes, eps)
    keep, safe = safe_reading(rl, rm, ref_boxes[..., 0].float(), logit_bound, cx_bound)
    mism = int(((rl != gl) & safe).sum())
    strings_equal = True
    n_chars = 0
    for b, idx in enumerate(keep):
        ref_s = rl[b, idx].tolist()
        gi = idx[torch.argsort(got_boxes[b, idx, 0].float(), stable=True)]
        got_s = gl[b, gi].tolist()
        n_chars += len(ref_s)
        strings_equal &= (ref_s == got_s)
    return dict(safe_frac=float(safe.float().mean()), label_mismatch_on_safe=mism, strings_equal=strings_equal, safe_chars=n_chars,
                chars_ref=int((rl >= 0).sum()), chars_got=int((gl >= 0).sum()), raw_label_agree=float((rl == gl).float().mean()))


# ---- n-gram re-scoring fixtures (shared by tests/golden/make_golden_ngram.py and the tests) -------------------------------------
def ngram_case(seed):
    """Seeded head outputs of ONE line whose argmax sequence looks like text: words of letters / digits / dashes separated by
    spaces and punctuation, blanks in between.  charset = the model's (label c <-> emission channel c + 1); ngram_charset = the
    n-gram side's table indexed by emission channel (index 0 = the CTC token)."""
    g = np.random.Generator(np.random.PCG64(4000 + seed))
    charset = list("abcdefgHIJ 0123-.,")
    ngram_charset = ["<ctc>"] + charset
    ignore = [ngram_charset.index(c) for c in " .,"]
    nq, C = 40, len(charset)
    logits = np.full((1, nq, C), -9.0, dtype=np.float32)
    cx = np.sort(g.uniform(0.02, 0.98, nq)).astype(np.float32)
    for q in range(nq):
        r = g.random()
        if r < 0.25:
            continue                                               # blank query
        c = int(g.integers(0, C)) if r < 0.9 else charset.index(" ")
        logits[0, q, c] = float(g.uniform(2.0, 8.0))
        if g.random() < 0.2:
            logits[0, q, int(g.integers(0, C))] = float(g.uniform(-1.0, 1.5))
    boxes = np.stack([cx, np.full(nq, 0.5, np.float32), np.full(nq, 0.02, np.float32), np.full(nq, 0.8, np.float32)], -1)[None]
    perm = g.permutation(nq)                                      # queries are not in reading order in the head output
    return ({"pred_logits": torch.from_numpy(logits[:, perm]), "pred_boxes": torch.from_numpy(boxes[:, perm].copy())},
            charset, ngram_charset, ignore)


def fake_ctc_decoder(ngram_charset):
    """A deterministic stand-in with torchaudio's ctc_decoder interface: greedy CTC collapse of the emissions, upper-cased, returned
    as hypothesis.words (a list of strings) -- enough to show WHICH spans were sent to the decoder and where its output lands."""
    class _H:
        def __init__(self, words):
            self.words = words

    def dec(em):
        lab = em[0].argmax(-1).tolist()
        out, prev = [], None
        for v in lab:
            if v != prev and v != 0:
                out.append(ngram_charset[v].upper())
            prev = v
        return [[_H(out)]]
    return dec
