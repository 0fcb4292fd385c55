"""Decoders and metrics of the reference's inference harness, batched on the GPU.

  convert_output_to_pred, NMS branch      <- evaluation.py:94-115   (scripts: --NMS 0.5 --TH 0.3)
  convert_output_to_pred, blank branch    <- evaluation.py:116-158  == SetCriterion.loss_CTC's
                                             blank construction (models/dino/dino.py:466-502)
                                             + engine.convert_output_to_pred (engine.py:511-530)
  CER / cumulative CER / normalisation    <- evaluation.py:296-334,430-450,517-529 ; engine.py:594-633
  WER, word splitting, gt normalisation   <- evaluation.py:358-428 ; engine.py:487-494,543-593
  per-character impact, WA, CR            <- evaluation.py:162-290
  the evaluation loop / CLI               <- evaluation.py:460-659  (dtlr_amd/eval_harness.py; `python -m dtlr_amd.evaluation`)

The reference decodes batch index 0 only (evaluation.py:154-155, batch size 1) with one `.item()`
device sync per character; here the whole batch is decoded on the device and ONE fixed-width record
per line (labels[nq] int32, length int32) crosses to the host -- the same record the data-parallel
driver all-gathers (dtlr_amd/dist.py).
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

from .dino import PostProcess, box_xyxy_to_cxcywh


def load_model(model, weights, device="cuda", new_class_embedding: bool = False, charset_size: Optional[int] = None,
               new_label_enc: bool = False, fix_enc_out_class: bool = False):
    """Checkpoint ingestion of the evaluation harness (evaluation.py:51-88), returning the model in eval mode on `device`.

    weights: path of a `checkpoint.pth` ({"model": state_dict}) or a state dict.  Without `new_class_embedding` the
    checkpoint is loaded as is (:54-59).  With it (HWDB / READ / cipher scripts) the class heads are first rebuilt to the
    dataset's charset size (:60-83): one Linear shared by the six decoder layers under `model.class_embed`, a separate
    bare Linear under `model.transformer.decoder.class_embed`, the two-stage head `enc_out_class_embed` unless
    `fix_enc_out_class`, and with `new_label_enc` the denoising label embedding (:84-85, unused at inference).
    charset_size None = take it from the checkpoint's `class_embed.0.weight` (what a matching charset has to equal)."""
    from . import weights as W
    sd = W.load_checkpoint_state_dict(weights) if isinstance(weights, (str, bytes)) or hasattr(weights, "__fspath__") else weights
    if new_class_embedding:
        features_dim = model.class_embed[0].weight.data.shape[1]
        n = int(charset_size) if charset_size is not None else W.num_classes_of(sd)
        new_class_embed = nn.Linear(features_dim, n)
        if not model.dec_pred_class_embed_share:
            raise NotImplementedError("load_model: the reference's head-resize flow only exists for dec_pred_class_embed_share "
                                      "(evaluation.py:75-79 leaves class_embed_layerlist undefined otherwise)")
        model.class_embed = nn.ModuleList([new_class_embed for _ in range(model.transformer.num_decoder_layers)])
        model.transformer.decoder.class_embed = nn.Linear(features_dim, n)
        if not fix_enc_out_class:
            model.transformer.enc_out_class_embed = nn.Linear(features_dim, n)
        if new_label_enc:
            model.label_enc = nn.Embedding(n + 1, features_dim)
    model.load_state_dict(sd)
    model.eval()
    return model.to(device)


@torch.no_grad()
def blank_probabilities(outputs: Dict[str, torch.Tensor], eps: float, scale: float = 1.0) -> torch.Tensor:
    """[B, nq, C+1] probabilities with the blank channel at index 0, queries sorted by box cx (HIP kernels: dtlr_blank_emissions)."""
    from . import ops
    return ops.blank_emissions(outputs["pred_logits"], outputs["pred_boxes"], eps, scale)


@torch.no_grad()
def decode_blank_records(outputs, eps: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Device-side blank/argmax decode (HIP kernel, one workgroup per line) -> (labels [B,nq] int32
    left-packed, -1 padded; lengths [B] int32).  No repeat collapse (engine.py:511-530, duplicate=False)."""
    from . import ops
    C = outputs["pred_logits"].shape[-1]
    return ops.decode_blank(outputs["pred_logits"], outputs["pred_boxes"], 0.03 / C if eps is None else eps)


def loss_ctc(outputs, target_labels: Sequence[Sequence[int]], eps: float = 0.003, filler: float = 1e-5) -> torch.Tensor:
    """Forward value of `SetCriterion.loss_CTC` (models/dino/dino.py:457-551) as engine.evaluate_CTC logs it
    (engine.py:381): HIP kernels (per-query sigmoid sums chip-wide, then one workgroup per line for the reading-order sort
    and the CTC alpha recursion over the 2 nq interleaved steps); the 'mean' reduction of nn.CTCLoss -- per-line NLL over
    max(target length, 1), averaged over the batch -- is applied here.  target_labels: per line, the label indices
    (charset positions, WITHOUT the +1 blank shift).  Returns a 0-d fp32 CUDA tensor."""
    from . import ops
    logits = outputs["pred_logits"]
    B = logits.shape[0]
    if len(target_labels) != B:
        raise ValueError(f"loss_ctc: {len(target_labels)} label sequences for a batch of {B}")
    lens = [len(t) for t in target_labels]
    Lmax = max(lens) if lens else 0
    tt = torch.zeros((B, max(Lmax, 1)), dtype=torch.int32)
    for i, t in enumerate(target_labels):
        if len(t):
            tt[i, : len(t)] = torch.as_tensor([int(v) for v in t], dtype=torch.int32) + 1
    tl = torch.tensor(lens, dtype=torch.int32)
    nll = ops.ctc_loss_interleaved(logits, outputs["pred_boxes"], tt.to(logits.device), tl.to(logits.device), Lmax, eps, filler)
    return (nll / tl.to(logits.device).clamp(min=1).float()).mean()


def evaluate_ctc_step(outputs, target_labels: Sequence[Sequence[int]]) -> Dict[str, float]:
    """One batch of engine.evaluate_CTC (engine.py:371-411): the CTC loss value and the summed per-line character error rate
    of the argmax decode that shares the loss's blank construction (eps = 0.003; engine.py:511-530, 544-568, 594-633).
    Returns {"loss_CTC", "cer_sum", "n"}: the caller accumulates cer_sum / n over the loader like engine.py:413-420."""
    loss = loss_ctc(outputs, target_labels)
    labels, lengths = decode_blank_records(outputs, eps=0.003)
    preds = records_to_lists(labels, lengths)
    cer = sum(character_error_rate(p, [int(v) for v in t]) for p, t in zip(preds, target_labels))
    return {"loss_CTC": float(loss.item()), "cer_sum": float(cer), "n": len(preds)}


def records_to_lists(labels: torch.Tensor, lengths: torch.Tensor) -> List[List[int]]:
    """decode records -> label lists.  A length of -1 marks a line whose logits were not finite (dtlr_decode_blank flags it on the device):
    on the float16 / split engines that is an activation beyond fp16's range (65504) -- raise instead of returning garbage."""
    lab, ln = labels.cpu().tolist(), lengths.cpu().tolist()
    bad = [i for i, n in enumerate(ln) if n < 0]
    if bad:
        from ._lib import DTLRError
        raise DTLRError(f"non-finite logits in line(s) {bad[:8]}{'...' if len(bad) > 8 else ''}: on the float16 / f32s engines an activation left fp16's "
                        "range (65504) -- run this checkpoint on the bfloat16 or the exact float32 engine")
    return [row[:n] for row, n in zip(lab, ln)]


def decode_blank(outputs, eps: Optional[float] = None) -> List[List[int]]:
    return records_to_lists(*decode_blank_records(outputs, eps))


@torch.no_grad()
def decode_nms(outputs, postprocessor: Optional[PostProcess] = None, TH: float = 0.3, NM: float = 0.5) -> List[List[int]]:
    """evaluation.py:94-115 for every line of the batch: PostProcess with num_select = #queries (900), NMS IoU NM on a (1,1)
    canvas, keep score > TH, order by box cx.  Top-k and the per-line filtering are batched device ops, the NMS one HIP launch
    for the whole batch (dtlr_nms); the only host transfers are the NMS counts and the final label lists."""
    pp = postprocessor or PostProcess()
    B, nq, _ = outputs["pred_logits"].shape
    pp.num_select, pp.nms_iou_threshold = min(900, nq) if nq < 900 else 900, NM
    dev = outputs["pred_logits"].device
    res = []
    for o in pp(outputs, torch.ones((B, 2), device=dev)):
        boxes = box_xyxy_to_cxcywh(o["boxes"])
        sel = o["scores"] > TH
        order = torch.sort(boxes[sel][:, 0], descending=False)[1]
        res.append(o["labels"].long()[sel][order])
    return [[int(i) for i in r.cpu().tolist()] for r in res]


def labels_to_string(labels: Sequence[int], charset: Sequence[str]) -> str:
    return "".join(charset[int(i)] for i in labels)


# ----------------------------------------------------------------------------------- metrics (host)
def levenshtein(s1, s2) -> int:
    """== editdistance.eval (evaluation.py:519,524); row-by-row DP."""
    if len(s1) < len(s2):
        s1, s2 = s2, s1
    if len(s2) == 0:
        return len(s1)
    prev = list(range(len(s2) + 1))
    for i, c1 in enumerate(s1):
        cur = [i + 1] + [0] * len(s2)
        for j, c2 in enumerate(s2):
            a, b, c = prev[j + 1] + 1, cur[j] + 1, prev[j] + (c1 != c2)
            cur[j + 1] = a if a < b and a < c else (b if b < c else c)
        prev = cur
    return prev[-1]


def character_error_rate(pred, gt) -> float:
    """engine.py:594-633: distance / max(len(gt),1); 1 if either side is empty."""
    if len(gt) == 0 or len(pred) == 0:
        return 1
    return levenshtein(pred, gt) / max(len(gt), 1)


def process_pred_string(s: str) -> str:
    """evaluation.py:430-450 (string normalisation before CER on IAM/RIMES/READ)."""
    for a, b in (("B B C", "BBC"), ("I T V", "ITV"), ("  ", " "), (" -", "-"), ("- ", "-"), (" .", "."), (" ,", ",")):
        s = s.replace(a, b)
    s = re.sub(r"(\d), (\d)", r"\1,\2", s)
    s = s.replace(" '", "'").replace("' ", "'")
    s = re.sub(r"(?<=\S)€(?=\S)", " € ", s)
    s = re.sub(r"(?<!\.)\.\.(?!\.)", ".", s)
    return s.replace(",,", ",")


def cumulative_cer(gt_strings: Sequence[str], pred_strings: Sequence[str], normalise: bool = True):
    """evaluation.py:517-529,547,653-656: running sum(dist)/sum(len); the reported number is the mean
    of that running series (reference quirk kept).  Returns (reported, series)."""
    dist, length, series = 0, 0, []
    for g, p in zip(gt_strings, pred_strings):
        if normalise:
            g, p = process_pred_string(g), process_pred_string(p)
        dist += levenshtein(g, p)
        length += len(g)
        series.append(dist / length)
    return (sum(series) / len(series) if series else 0.0), series


def word_error_rate(predicted_words, gt_words) -> float:
    """evaluation.py:358-396: word-level edit distance / max(#gt words, 1).  The reference's loop passes (gt_split, pred_split)
    (evaluation.py:533-535, 546-549), so the figure it reports is normalised by the number of PREDICTED words; the harness
    here calls it the same way."""
    return levenshtein(predicted_words, gt_words) / max(len(gt_words), 1)


def split_labels_into_words(labels: Sequence[int], charset: Sequence[str]) -> List[List[int]]:
    """evaluation.py:400-412: cut a label sequence at the charset's space; no empty words."""
    space = list(charset).index(" ")
    words: List[List[int]] = [[]]
    for lab in labels:
        if lab == space:
            if words[-1]:
                words.append([])
        else:
            words[-1].append(lab)
    return words if words[-1] else words[:-1]


_GT_RULES = (("B B C", "BBC"), ("I T V", "ITV"), (" -", "-"), ("- ", "-"), (" -", "-"), ("- ", "-"), (" .", "."), (" ,", ","),
             (" '", "'"), ("' ", "'"))


def process_gt_string(s: str) -> str:
    """evaluation.py:414-428 (ground-truth normalisation; unlike process_pred_string it does not collapse double blanks or
    repeated punctuation)."""
    for a, b in _GT_RULES:
        s = s.replace(a, b)
    s = re.sub(r"(\d), (\d)", r"\1,\2", s)
    return re.sub(r"(?<=\S)€(?=\S)", " € ", s)


def character_error_rate_with_impact(pred: Sequence[int], gt: Sequence[int], impact: Dict[int, int]):
    """evaluation.py:162-210 -> (cer, impact, div).  `impact[c]` grows, for every predicted character c, by the number of
    ground-truth characters that differ from it (what the reference's bookkeeping inside its DP loop amounts to); written to
    dict_char.json by the harness.  An empty ground truth raises (the reference fails on it too)."""
    if len(gt) == 0:
        raise ValueError("character_error_rate_with_impact: empty ground truth")
    counts: Dict[int, int] = {}
    for g_ in gt:
        counts[int(g_)] = counts.get(int(g_), 0) + 1
    for p_ in pred:
        n = len(gt) - counts.get(int(p_), 0)
        if n:
            impact[int(p_)] = impact.get(int(p_), 0) + n
    return levenshtein(pred, gt) / len(gt), impact, len(gt)


def compute_wa(gt: Sequence[int], pred: Sequence[int]) -> float:
    """evaluation.py:212-238 (cipher "word accuracy"): matching positions over max(len(gt), 1)."""
    return sum(1 for a, b in zip(gt, pred) if a == b) / max(len(gt), 1)


def compute_edit_operations(s1: Sequence, s2: Sequence) -> Tuple[int, int, int]:
    """evaluation.py:239-281 -> (insertions, deletions, substitutions) of the alignment the reference's backtrace picks
    (substitution before deletion before insertion)."""
    m, n = len(s1), len(s2)
    rows = [list(range(n + 1))]
    for i in range(1, m + 1):
        prev, cur = rows[-1], [i] + [0] * n
        for j in range(1, n + 1):
            cur[j] = prev[j - 1] if s1[i - 1] == s2[j - 1] else 1 + min(prev[j], cur[j - 1], prev[j - 1])
        rows.append(cur)
    i, j, ins, dele, sub = m, n, 0, 0, 0
    while i and j:
        here = rows[i][j]
        if s1[i - 1] == s2[j - 1]:
            i, j = i - 1, j - 1
        elif here == rows[i - 1][j - 1] + 1:
            sub, i, j = sub + 1, i - 1, j - 1
        elif here == rows[i - 1][j] + 1:
            dele, i = dele + 1, i - 1
        else:
            ins, j = ins + 1, j - 1
    return ins + j, dele + i, sub


def compute_cr(gt: Sequence[int], pred: Sequence[int]) -> float:
    """evaluation.py:283-290 (Chinese correct rate): (len(gt) - deletions - substitutions) / len(gt)."""
    _, dele, sub = compute_edit_operations(gt, pred)
    return (len(gt) - dele - sub) / len(gt)


_WER_PUNCT = re.compile(r"""([\[\]{}/\()"'&+*=<>?.;:,!\-—_€#%°])""")


def format_string_for_wer(s: str) -> List[str]:
    """engine.py:487-494: every punctuation mark is a word of its own; blanks and line breaks collapse."""
    s = _WER_PUNCT.sub(r" \1 ", s)
    return re.sub(r"[ \n]+", " ", s).strip().split(" ")


def compute_wer(pred_labels: Sequence[Sequence[int]], target_labels: Sequence[Sequence[int]], charset: Sequence,
                mode_chr: bool = True) -> Tuple[float, float]:
    """engine.py:543-593 on decoded label sequences (its duplicate=False path): (sum of per-line WER, sum of per-line CER),
    WER = word edit distance of the formatted strings / number of ground-truth words, '¬' removed."""
    wer = cer = 0.0
    for pred, tgt in zip(pred_labels, target_labels):
        tgt = [int(t) for t in tgt]
        cer += character_error_rate(list(pred), tgt)
        conv = (lambda c: c) if mode_chr else (lambda c: chr(int(c)))
        gt_words = format_string_for_wer("".join(conv(charset[t]) for t in tgt).replace("¬", ""))
        pr_words = format_string_for_wer("".join(conv(charset[int(p)]) for p in pred).replace("¬", ""))
        wer += levenshtein(gt_words, pr_words) / len(gt_words)
    return wer, cer


if __name__ == "__main__":
    from .eval_harness import main
    main()
