"""N-gram re-scoring of the recogniser's output (reference: ngram/prediction_helpers.py:5-224).

The reference turns the detector output into CTC-style emissions (`get_new_pred_logits`: queries sorted by box cx, sigmoid,
blank channel with eps = 0.003), cuts the line at the characters it never rescoses (`indices_to_ignore`: space, punctuation ...),
and sends every word's emissions through torchaudio's lexicon CTC beam decoder with a KenLM character n-gram
(`torchaudio.models.decoder.ctc_decoder`, prediction_helpers.py:72-90).

  * the emission tensor is built on the device by dtlr_blank_emissions (csrc/decode.hip: the blank decoder's per-query sigmoid sums and
    reading-order sort, then one wave per output row; `evaluation.blank_probabilities`) -- GPU only, like every other operator;
  * the word-splitting / re-assembly logic is restated here (host logic on one label row per line);
  * torchaudio / flashlight-text / KenLM are third-party packages that are neither in the reference tree nor installed here: the
    decoder is a CALLABLE with torchaudio's interface (`decoder(emissions [1,T,V]) -> [[hypothesis]]`, hypothesis.words), so a
    user holding those packages passes `torchaudio.models.decoder.ctc_decoder(...)` unchanged.  `LexiconCTCDecoder` below is a
    small self-contained stand-in with the same interface -- CTC prefix beam search constrained to a lexicon, scored with an ARPA
    n-gram (`ArpaLM`) -- restating the published algorithm.  PARITY UNPINNED: nothing in this container can run torchaudio's decoder or
    KenLM, so `LexiconCTCDecoder` / `ArpaLM` are checked only against hand-computed cases; what IS pinned to the reference (G8: vectors
    produced by its own function bodies) is everything around the decoder: emissions, span selection, re-assembly.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import evaluation as E


@torch.no_grad()
def get_new_pred_logits(output: Dict[str, torch.Tensor], multiply_pred_logits_by: float = 1.0) -> torch.Tensor:
    """prediction_helpers.py:5-46: [B, nq, C+1] emissions, queries in reading order, blank channel first (eps 0.003).  With the
    default multiplier this is exactly the evaluation loss's blank construction (models/dino/dino.py:466-502)."""
    return E.blank_probabilities(output, 0.003, float(multiply_pred_logits_by))


def _first_non0(labels: Sequence[int]) -> int:
    """prediction_helpers.py:117-121 (raises like the reference on an empty list: the caller skips that span)."""
    e = None
    for e in labels:
        if e > 0:
            break
    if e is None:
        raise UnboundLocalError("empty span")
    return e


def get_input_split_indices(model_labels: Sequence[int], ngram_charset: Sequence[str], indices_to_ignore: Sequence[int],
                            no_uppercase_words: bool = True, no_digits: bool = False, no_dash: bool = True) -> Tuple[List[int], List[int]]:
    """prediction_helpers.py:124-173 on one row of argmax labels (0 = blank): positions of the never-rescored characters, and the
    subset of spans that may be sent to the n-gram decoder (not starting with an upper-case letter / digit, no dash inside)."""
    labels = [int(v) for v in model_labels]
    ignore = set(int(i) for i in indices_to_ignore)
    split = [-1] + [i for i, v in enumerate(labels) if v in ignore] + [len(labels)]
    if not (no_uppercase_words or no_digits):
        return split, split
    clean: List[int] = []
    for i in range(len(split) - 1):
        try:
            first = _first_non0(labels[split[i] + 1: split[i + 1] - 1])
        except UnboundLocalError:
            continue
        if first == 0:
            continue
        if no_uppercase_words and ngram_charset[first].isupper():
            continue
        if no_digits and ngram_charset[first].isdigit():
            continue
        elif no_dash and (list(ngram_charset).index("-") in labels[split[i] + 1: split[i + 1]]):
            continue
        else:
            clean.append(split[i])
    clean.append(len(labels))
    return split, clean


def get_word_per_word_pred(new_pred_logits: torch.Tensor, ctc_decoder: Callable, indices_to_ignore: Sequence[int], charset: Sequence[str]) -> str:
    """prediction_helpers.py:49-74: every span between never-rescored characters goes through the decoder; the separators are copied
    from the argmax (charset[label - 1])."""
    row = new_pred_logits[0]
    labels = row.argmax(-1).tolist()
    ignore = set(int(i) for i in indices_to_ignore)
    split = [-1] + [i for i, v in enumerate(labels) if v in ignore] + [len(labels)]
    chars: List[str] = []
    for i in range(len(split) - 1):
        if split[i] < split[i + 1] - 1:
            word = row[split[i] + 1: split[i + 1]][None, :, :]
            chars += ctc_decoder(word.cpu())[0][0].words
        if split[i + 1] < len(labels):
            chars += charset[labels[split[i + 1]] - 1]
    return "".join(chars)


def get_word_per_word_pred_2(new_pred_logits: torch.Tensor, ctc_decoder: Callable, indices_to_ignore: Sequence[int],
                             ngram_charset: Sequence[str], no_uppercase_words: bool, no_digits: bool, no_dash: bool) -> str:
    """prediction_helpers.py:176-224: like the above, but spans that must not be rescored keep their argmax characters."""
    row = new_pred_logits[0]
    labels = row.argmax(-1).tolist()
    split, clean = get_input_split_indices(labels, ngram_charset, indices_to_ignore, no_uppercase_words, no_digits, no_dash)
    chars: List[str] = []
    max_added = -1
    inner, head = set(split[1:]), set(split[:-1])
    clean_set = set(clean)
    for i in range(len(split) - 1):
        a, b = split[i], split[i + 1]
        if a in inner and a > max_added:
            chars += ngram_charset[labels[a]]
            max_added = a
        if a < b and a in clean_set:
            chars += ctc_decoder(row[a + 1: b][None, :, :].cpu())[0][0].words
            max_added = max(b - 1, max_added)
        else:
            chars += [ngram_charset[v] for v in labels[a + 1: b] if v > 0]
            max_added = max(b - 1, max_added)
        if b in head and b > max_added:
            chars += ngram_charset[labels[b]]
            max_added = b
    return "".join(chars)


@torch.no_grad()
def get_ngram_prediction(outputs, ctc_decoder: Callable, indices_to_ignore, charset, ngram_charset, per_word_ngram: bool = True,
                         no_uppercase_words: bool = False, no_digits: bool = False, no_dash: bool = True) -> str:
    """prediction_helpers.py:93-114 for ONE line (`outputs` with batch size 1), the decoder passed in instead of built from a config."""
    emissions = get_new_pred_logits(outputs).cpu()          # HIP kernels on the device; the word assembly below is host logic (one copy per line)
    if per_word_ngram and (no_uppercase_words or no_digits):
        return get_word_per_word_pred_2(emissions, ctc_decoder, indices_to_ignore, ngram_charset, no_uppercase_words, no_digits, no_dash)
    if per_word_ngram:
        return get_word_per_word_pred(emissions, ctc_decoder, indices_to_ignore, charset)
    raise NotImplementedError("no test support for full sentence n-gram for now")      # as the reference (:108)


# ----------------------------------------------------------------------------------------------------------------------------
class ArpaLM:
    """Back-off n-gram language model read from an ARPA text file (the format KenLM's `lmplz` writes before `build_binary`;
    the reference trains character 5-grams, ngram/train_n_gram.sh).  log10 scores; Katz back-off."""

    def __init__(self, path: str):
        self.order = 0
        self.grams: Dict[Tuple[str, ...], Tuple[float, float]] = {}
        section = 0
        with open(path, encoding="utf-8") as f:
            for line in f:
                line = line.strip()
                if not line or line == "\\data\\" or line.startswith("ngram "):
                    continue
                if line.startswith("\\") and line.endswith("-grams:"):
                    section = int(line[1:line.index("-")])
                    self.order = max(self.order, section)
                    continue
                if line == "\\end\\":
                    break
                parts = line.split("\t")
                words = tuple(parts[1].split(" "))
                self.grams[words] = (float(parts[0]), float(parts[2]) if len(parts) > 2 else 0.0)

    def score(self, context: Tuple[str, ...], word: str) -> float:
        """log10 P(word | context) with back-off."""
        context = context[-(self.order - 1):] if self.order > 1 else ()
        while True:
            hit = self.grams.get(context + (word,))
            if hit is not None:
                return hit[0]
            if not context:
                return self.grams.get(("<unk>",), (-10.0, 0.0))[0]
            bo = self.grams.get(context, (0.0, 0.0))[1]
            return bo + self.score(context[1:], word)


class _Hypothesis:
    def __init__(self, words, score):
        self.words, self.score = words, score


class LexiconCTCDecoder:
    """CTC prefix beam search over a lexicon of token sequences with an optional n-gram LM, with torchaudio's call interface:
    decoder(emissions [B,T,V] probabilities or log-probabilities) -> [[hypothesis]] with hypothesis.words (list of lexicon words).
    tokens: list of V token strings (index = emission channel); lexicon: {word: [token, ...]}; blank / word boundary tokens named.
    A hypothesis is a sequence of complete lexicon words; within a word the search follows the lexicon trie."""

    def __init__(self, tokens: Sequence[str], lexicon: Dict[str, Sequence[str]], lm: Optional[ArpaLM] = None, lm_weight: float = 0.0,
                 blank_token: str = "<ctc>", sil_token: str = "<space>", beam_size: int = 50, log_probs: bool = False):
        self.tokens = list(tokens)
        self.blank = self.tokens.index(blank_token)
        self.sil = self.tokens.index(sil_token) if sil_token in self.tokens else -1
        self.lm, self.lm_weight, self.beam, self.log_probs = lm, lm_weight, beam_size, log_probs
        self.trie: dict = {}
        for word, spelling in lexicon.items():
            node = self.trie
            for t in spelling:
                node = node.setdefault(self.tokens.index(t), {})
            node.setdefault(-1, []).append(word)

    def _decode_one(self, em: torch.Tensor) -> List[_Hypothesis]:
        lp = em.double() if self.log_probs else torch.log(em.double().clamp_min(1e-30))
        T = lp.shape[0]
        NEG = -1e30
        # beam entry key: (words tuple, trie path tuple) -> (log p ending in blank, log p ending in non-blank, lm score)
        beams = {((), ()): (0.0, NEG, 0.0)}

        def lse(a, b):
            if a < b:
                a, b = b, a
            return a if b <= NEG / 2 else a + math.log1p(math.exp(b - a))

        for t in range(T):
            nxt: Dict[tuple, list] = {}

            def add(key, pb, pnb, lm):
                cur = nxt.get(key)
                if cur is None:
                    nxt[key] = [pb, pnb, lm]
                else:
                    cur[0], cur[1] = lse(cur[0], pb), lse(cur[1], pnb)
            row = lp[t].tolist()
            for (words, path), (pb, pnb, lm) in beams.items():
                tot = lse(pb, pnb)
                add((words, path), tot + row[self.blank], NEG, lm)                       # emit blank
                node = self.trie
                for tok in path:
                    node = node[tok]
                if path:                                                                  # repeat the last token (CTC collapse)
                    add((words, path), NEG, pnb + row[path[-1]], lm)
                for tok, child in node.items():
                    if tok < 0:
                        continue
                    p_new = (pb if (path and tok == path[-1]) else tot) + row[tok]         # a repeated char needs a blank in between
                    add((words, path + (tok,)), NEG, p_new, lm)
                if path and -1 in node:                                                   # word complete: close it (no emission consumed)
                    for w in node[-1]:
                        lm_new = lm + (self.lm_weight * self.lm.score(tuple(words), w) * math.log(10.0) if self.lm else 0.0)
                        add((words + (w,), ()), pb, pnb, lm_new)
            ranked = sorted(nxt.items(), key=lambda kv: -(lse(kv[1][0], kv[1][1]) + kv[1][2]))[: self.beam]
            beams = {k: tuple(v) for k, v in ranked}
        final = []
        for (words, path), (pb, pnb, lm) in beams.items():
            node = self.trie
            for tok in path:
                node = node[tok]
            if path and -1 in node:
                for w in node[-1]:
                    lm2 = lm + (self.lm_weight * self.lm.score(tuple(words), w) * math.log(10.0) if self.lm else 0.0)
                    final.append(_Hypothesis(list(words) + [w], lse(pb, pnb) + lm2))
            elif not path:
                final.append(_Hypothesis(list(words), lse(pb, pnb) + lm))
        final.sort(key=lambda h: -h.score)
        return final or [_Hypothesis([], NEG)]

    def __call__(self, emissions: torch.Tensor) -> List[List[_Hypothesis]]:
        return [self._decode_one(e) for e in emissions]
