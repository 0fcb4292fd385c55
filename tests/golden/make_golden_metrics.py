"""Golden vectors for the host-side metrics (authoring container only): lifts the metric FUNCTIONS out of the reference's
evaluation.py / engine.py with `ast` (both files run argparse / import datasets at module level and cannot be imported), executes
only those definitions, and stores their outputs on seeded inputs as data.

    python -m tests.golden.make_golden_metrics        # writes tests/golden/g6_metrics.json

`editdistance` (an un-vendored C++ Levenshtein, requirements.txt:14) is replaced by the reference's own pure-Python Levenshtein
(the inner function of evaluation.character_error_rate): plain edit distance on any two sequences.
"""
from __future__ import annotations

import ast
import json
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

EVAL_FUNCS = ["word_error_rate", "split_labels_into_words", "process_gt_string", "process_pred_string", "character_error_rate_with_impact",
              "compute_WA", "compute_edit_operations", "compute_CR", "character_error_rate"]
ENGINE_FUNCS = ["format_string_for_wer", "edit_wer_from_formatted_split_text", "convert_output_to_pred", "compute_wer", "character_error_rate"]


def lift(path, names, extra):
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in body} == set(names), set(names) - {n.name for n in body}
    ns = {"torch": torch, "re": re, "np": np}
    ns.update(extra)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def main():
    ev = lift(os.path.join(REF, "evaluation.py"), EVAL_FUNCS, {})

    def lev(a, b):                     # the reference's own Levenshtein, via its CER function (distance = cer * max(len(gt), 1))
        a, b = list(a), list(b)
        return round(ev["character_error_rate"](a, b) * max(len(b), 1))

    ed = types.SimpleNamespace(eval=lev)
    en = lift(os.path.join(REF, "engine.py"), ENGINE_FUNCS, {"editdistance": ed})
    charset = json.load(open(os.path.join(REF, "datasets", "default_charset.json")))
    space = charset.index(" ")
    g = np.random.Generator(np.random.PCG64(77))
    out = {"charset_len": len(charset), "space_index": space, "cases": []}
    for k in range(24):
        n = int(g.integers(0, 40))
        gt = g.integers(0, 60, max(n, 1)).tolist()
        for pos in g.integers(0, len(gt), max(1, len(gt) // 5)).tolist():
            gt[pos] = space
        pred = list(gt)
        for _ in range(int(g.integers(0, 8))):                         # random edits
            op, pos = int(g.integers(0, 3)), int(g.integers(0, max(len(pred), 1)))
            if op == 0 and pred:
                pred[pos] = int(g.integers(0, 60))
            elif op == 1:
                pred.insert(pos, int(g.integers(0, 60)) if g.random() < 0.8 else space)
            elif pred:
                pred.pop(pos)
        if k == 3:
            pred = []
        c = {"gt": gt, "pred": pred}
        gw, pw = ev["split_labels_into_words"](gt, charset), ev["split_labels_into_words"](pred, charset)
        c["gt_words"], c["pred_words"] = gw, pw
        c["wer_as_called"] = ev["word_error_rate"](gw, pw)             # evaluation.py:533-535 argument order
        c["wer_declared"] = ev["word_error_rate"](pw, gw)
        c["wa"] = ev["compute_WA"](gt, pred)
        c["edit_ops"] = list(ev["compute_edit_operations"](gt, pred))
        c["cr"] = ev["compute_CR"](gt, pred)
        c["cer_eval"] = ev["character_error_rate"](pred, gt)
        c["cer_engine"] = en["character_error_rate"](pred, gt)
        if pred:
            cer, impact, div = ev["character_error_rate_with_impact"](pred, gt, {})
            c["impact"] = {"cer": cer, "div": div, "dict": {str(a): b for a, b in sorted(impact.items())}}
        # engine.compute_wer on "new_pred_logits" that decode to exactly `pred` (argmax = label + 1, blanks elsewhere)
        L = len(pred) + 3
        newp = torch.zeros((1, L, len(charset) + 1))
        newp[0, :, 0] = 1.0
        for i, lab in enumerate(pred):
            newp[0, i + 1, 0] = 0.0
            newp[0, i + 1, lab + 1] = 1.0
        wer, cer2 = en["compute_wer"]({"pred_logits": newp}, [{"labels": torch.tensor(gt)}], charset, newp, mode_chr=True)
        c["engine_wer"], c["engine_cer"] = float(wer), float(cer2)
        out["cases"].append(c)
    strings = ["The B B C and I T V said - well , 'tis 1, 2 . . . done", "price 5€3 and a€b ,, twice .. or ... thrice",
               "spaces  doubled  here - and- there -x", "quote ' a ' b", "10, 000 people , 3, 5", "plain text"]
    out["strings"] = [{"s": s_, "gt": ev["process_gt_string"](s_), "pred": ev["process_pred_string"](s_),
                       "words": en["format_string_for_wer"](s_)} for s_ in strings]
    with open(os.path.join(HERE, "g6_metrics.json"), "w") as f:
        json.dump(out, f, indent=0, ensure_ascii=False)
    print("g6 written:", len(out["cases"]), "cases")


if __name__ == "__main__":
    assert os.path.isdir(REF), "needs /root/reference"
    main()
