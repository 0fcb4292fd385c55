"""Decode-level comparison of two head outputs of the same queries (TEST INFRASTRUCTURE, like everything under oracle/: used by
tests/ and by bench.py's `cer_vs_oracle` checker leg, never by the product path).

`query_decisions` restates the per-query view of the reference's blank / argmax decoder (/root/reference/evaluation.py:116-158,
models/dino/dino.py:466-502) together with a decision MARGIN in logit units; `compare_decoded` is the round-2/3 gate (equality on the
"safe" queries only); `tie_aware_compare` (round 4) accounts for EVERY query: a label may differ only where the oracle's margin is
below twice the measured logit error, two characters may trade places in the reading order only if their oracle cx differ by less
than twice the measured cx error, and the count of differences explained by neither must be zero."""
import torch


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




def _levenshtein(a, b):
    """plain edit distance on label lists (/root/reference/evaluation.py:309-326 is the reference's own pure-Python version)."""
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def decode_order(labels, cx):
    """reading-order label lists of the non-blank queries (stable sort by cx: evaluation.py:122-127)."""
    out = []
    for b in range(labels.shape[0]):
        idx = torch.nonzero(labels[b] >= 0).flatten()
        idx = idx[torch.argsort(cx[b, idx], stable=True)]
        out.append(labels[b, idx].tolist())
    return out


def tie_aware_compare(ref_logits, ref_boxes, got_logits, got_boxes, eps, logit_err, cx_err):
    """Every query counted.  Returns a dict:
      chars_ref / edit_distance / cer_all_queries : the decoded strings of ALL queries, reference vs candidate
      label_flips, label_flips_unexplained        : queries whose decision differs; unexplained = reference margin >= 2 * logit_err
      order_swaps, order_swaps_unexplained        : pairs of printing queries whose reading order differs; unexplained = their
                                                    reference cx differ by >= 2 * cx_err
      unexplained                                 : the sum of the two -- must be 0 for a candidate whose only difference from the
                                                    reference is the measured (logit_err, cx_err)
      min_gap_px_2048                             : the smallest cx gap between neighbouring reference characters, in pixels of a
                                                    2048-wide canvas (the synthetic heads place characters at random: sub-pixel
                                                    neighbours exist, which no real line has)"""
    rl, rm = query_decisions(ref_logits, ref_boxes, eps)
    gl, _ = query_decisions(got_logits, got_boxes, eps)
    rcx, gcx = ref_boxes[..., 0].float(), got_boxes[..., 0].float()
    flips = rl != gl
    flips_unexp = flips & (rm >= 2 * logit_err)
    swaps = swaps_unexp = 0
    min_gap = float("inf")
    for b in range(rl.shape[0]):
        idx = torch.nonzero((rl[b] >= 0) | (gl[b] >= 0)).flatten()
        if len(idx) < 2:
            continue
        n = len(idx)
        rr = torch.empty(n, dtype=torch.long)
        rr[torch.argsort(rcx[b, idx], stable=True)] = torch.arange(n)
        rg = torch.empty(n, dtype=torch.long)
        rg[torch.argsort(gcx[b, idx], stable=True)] = torch.arange(n)
        disc = (rr[:, None] < rr[None, :]) & (rg[:, None] > rg[None, :])
        far = (rcx[b, idx][:, None] - rcx[b, idx][None, :]).abs() >= 2 * cx_err
        swaps += int(disc.sum())
        swaps_unexp += int((disc & far).sum())
        c = torch.sort(rcx[b, torch.nonzero(rl[b] >= 0).flatten()])[0]
        if len(c) > 1:
            min_gap = min(min_gap, float((c[1:] - c[:-1]).min()))
    a, g = decode_order(rl, rcx), decode_order(gl, gcx)
    dist = sum(_levenshtein(x, y) for x, y in zip(a, g))
    n_chars = sum(len(x) for x in a)
    return dict(chars_ref=n_chars, edit_distance=dist, cer_all_queries=dist / max(n_chars, 1),
                label_flips=int(flips.sum()), label_flips_unexplained=int(flips_unexp.sum()),
                order_swaps=swaps, order_swaps_unexplained=swaps_unexp, unexplained=int(flips_unexp.sum()) + swaps_unexp,
                min_gap_px_2048=(min_gap * 2048 if min_gap < float("inf") else None))
