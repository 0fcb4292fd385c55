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


# ---- margin-aware comparison of decoded outputs: moved to oracle/compare.py (checker code shared with bench.py's parity leg) -------
from oracle.compare import compare_decoded, query_decisions, safe_reading, tie_aware_compare  # noqa: E402,F401


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
