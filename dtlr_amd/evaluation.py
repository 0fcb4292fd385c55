"""Decoders and metrics of the reference's inference harness, batched on the GPU.

  convert_output_to_pred, NMS branch      <- evaluation.py:94-115   (scripts: --NMS 0.5 --TH 0.3)
  convert_output_to_pred, blank branch    <- evaluation.py:116-158  == SetCriterion.loss_CTC's
                                             blank construction (models/dino/dino.py:466-502)
                                             + engine.convert_output_to_pred (engine.py:511-530)
  CER / cumulative CER / normalisation    <- evaluation.py:296-334,430-450,517-529 ; engine.py:594-633

The reference decodes batch index 0 only (evaluation.py:154-155, batch size 1) with one `.item()`
device sync per character; here the whole batch is decoded on the device and ONE fixed-width record
per line (labels[nq] int32, length int32) crosses to the host -- the same record the data-parallel
driver all-gathers (dtlr_amd/dist.py).
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .dino import PostProcess, box_xyxy_to_cxcywh


@torch.no_grad()
def blank_probabilities(outputs: Dict[str, torch.Tensor], eps: float) -> torch.Tensor:
    """[B, nq, C+1] probabilities with the blank channel at index 0, queries sorted by box cx."""
    logits, boxes = outputs["pred_logits"].float(), outputs["pred_boxes"].float()
    _, idx = torch.sort(boxes[:, :, 0])
    p = torch.gather(logits, 1, idx.unsqueeze(-1).expand(-1, -1, logits.shape[-1])).sigmoid()
    s = p.sum(-1, keepdim=True)
    low = s < 1 - eps
    blank = torch.where(low, 1 - s, torch.full_like(s, eps))
    cls = torch.where(low, p, (1 - eps) * p / s)
    return torch.cat([blank, cls], -1)


@torch.no_grad()
def decode_blank_records(outputs, eps: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Device-side blank/argmax decode (HIP kernel, one workgroup per line) -> (labels [B,nq] int32
    left-packed, -1 padded; lengths [B] int32).  No repeat collapse (engine.py:511-530, duplicate=False)."""
    from . import ops
    C = outputs["pred_logits"].shape[-1]
    return ops.decode_blank(outputs["pred_logits"], outputs["pred_boxes"], 0.03 / C if eps is None else eps)


def records_to_lists(labels: torch.Tensor, lengths: torch.Tensor) -> List[List[int]]:
    lab, ln = labels.cpu().tolist(), lengths.cpu().tolist()
    return [row[:n] for row, n in zip(lab, ln)]


def decode_blank(outputs, eps: Optional[float] = None) -> List[List[int]]:
    return records_to_lists(*decode_blank_records(outputs, eps))


@torch.no_grad()
def decode_nms(outputs, postprocessor: Optional[PostProcess] = None, TH: float = 0.3, NM: float = 0.5) -> List[List[int]]:
    """evaluation.py:94-115 per line: PostProcess with num_select = #queries(900), NMS IoU NM on a
    (1,1) canvas, keep score > TH, order by box cx."""
    pp = postprocessor or PostProcess()
    B, nq, _ = outputs["pred_logits"].shape
    pp.num_select, pp.nms_iou_threshold = min(900, nq) if nq < 900 else 900, NM
    res = []
    dev = outputs["pred_logits"].device
    for b in range(B):
        one = {"pred_logits": outputs["pred_logits"][b:b + 1], "pred_boxes": outputs["pred_boxes"][b:b + 1]}
        o = pp(one, torch.tensor([[1.0, 1.0]], device=dev))[0]
        boxes = box_xyxy_to_cxcywh(o["boxes"])
        sel = o["scores"] > TH
        order = torch.sort(boxes[sel][:, 0], descending=False)[1]
        res.append([int(i) for i in o["labels"].long()[sel][order].cpu().tolist()])
    return res


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
