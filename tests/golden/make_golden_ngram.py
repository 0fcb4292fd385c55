"""Golden vectors for the n-gram re-scoring glue (authoring container only): the reference's functions are lifted out of
ngram/prediction_helpers.py with `ast` (the module imports torchaudio at the top and moves tensors to "cuda": string constants
"cuda" are rewritten to "cpu" in the syntax tree), run with a FAKE deterministic ctc_decoder, and their outputs stored as data.

    python -m tests.golden.make_golden_ngram          # writes tests/golden/g8_ngram.json
"""
from __future__ import annotations

import ast
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.util import fake_ctc_decoder, ngram_case     # noqa: E402

REF = "/root/reference/ngram/prediction_helpers.py"
FUNCS = ["get_new_pred_logits", "get_word_per_word_pred", "get_first_non_0_charac", "get_input_split_indices", "get_word_per_word_pred_2"]


class _Cpu(ast.NodeTransformer):
    def visit_Constant(self, node):
        return ast.copy_location(ast.Constant("cpu"), node) if node.value == "cuda" else node


def main():
    tree = _Cpu().visit(ast.parse(open(REF).read()))
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in FUNCS]
    ns = {"torch": torch}
    exec(compile(ast.fix_missing_locations(ast.Module(body=body, type_ignores=[])), REF, "exec"), ns)
    out = {"cases": []}
    for seed in range(10):
        outputs, charset, ngram_charset, ignore = ngram_case(seed)
        new = ns["get_new_pred_logits"](outputs)
        rec = {"seed": seed, "new_sum": float(new.double().sum()), "new_argmax": new[0].argmax(-1).tolist()}
        rec["word_per_word"] = ns["get_word_per_word_pred"](new, fake_ctc_decoder(ngram_charset), ignore, charset)
        for k, (up, dg, ds) in enumerate(((True, False, True), (False, True, True), (True, True, False))):
            cfg = types.SimpleNamespace(no_uppercase_words=up, no_digits=dg, no_dash=ds)
            rec[f"split_{k}"] = [list(map(int, v)) for v in ns["get_input_split_indices"](new, ngram_charset, ignore, up, dg, ds)]
            rec[f"word_per_word_2_{k}"] = ns["get_word_per_word_pred_2"](new, fake_ctc_decoder(ngram_charset), ignore, ngram_charset, cfg)
        out["cases"].append(rec)
    with open(os.path.join(HERE, "g8_ngram.json"), "w") as f:
        json.dump(out, f, ensure_ascii=False)
    print("g8 written")


if __name__ == "__main__":
    main()
