"""The evaluation loop of the reference's harness (evaluation.py:460-659) on the MI355X engine:

    python -m dtlr_amd.evaluation --config latin --weights checkpoint.pth --images DIR --labels labels.pkl --mode test \
           --dataset IAM --NMS 0.5 --TH 0.3                       # scripts/evaluating/IAM.sh
    python -m torch.distributed.run --nproc-per-node 8 -m dtlr_amd.evaluation ...      # data-parallel over the node

checkpoint + folder of line images + labels  ->  preprocess (device) -> forward -> decode -> CER / WER / AR / CR / WA, and the
reference's output files under <out>/<dataset>/: cer_list.npy, dict_char.json, list_preds.txt, list_gt.txt,
cer_TH_{TH}_NMS_{NM}.txt (the character-impact histogram PNG is visualisation and is not produced).

What differs from the reference loop, and why the numbers do not:
  * the reference forwards ONE image at a time (`model(image[None])`, evaluation.py:499).  Here lines whose resized size is
    identical are batched together WITHOUT padding (`--batching exact`, default): the per-line arithmetic is exactly the
    bs = 1 arithmetic, only the launch is shared.  `--batching padded` pads mixed sizes into one canvas (faster, but a padded
    line is not bit-identical to the same line alone: the backbone sees the canvas's zero padding instead of its own border);
  * lines are sharded over the ranks of a torch.distributed job (contiguous shards of the size-sorted list), decoded records
    are all-gathered, rank 0 computes the metrics in dataset order -- the running CER series (the figure the reference reports
    is the MEAN of the running sum(dist)/sum(len) series, evaluation.py:521-529,547) is order dependent;
  * a line whose forward raises is skipped by the reference (evaluation.py:498-504: message, `continue`, the line enters no
    metric).  Same here: a failing batch is retried line by line, a line that still fails (or whose image file cannot be read)
    is reported on stderr and left out of every list -- one bad image does not end a 2915-line run, and does not take its
    batch neighbours with it;
  * every rank reads only the image HEADERS of the whole set (sizes, for the batch plan) and decodes only the lines of its own
    shard -- not N copies of the dataset in host memory.
"""
from __future__ import annotations

import argparse
import json
import os
import pickle
import sys
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import dist as ddist
from . import evaluation as E
from .config import DTLRConfig
from .transforms import EVAL_MAX_SIZE, EVAL_SIZE, EvalTransform, get_size_with_aspect_ratio

HERE = os.path.dirname(os.path.abspath(__file__))
CER_DATASETS = ("IAM", "RIMES", "READ")          # evaluation.py:510,521: string-level cumulative CER + WER


def load_charset(path: Optional[str]) -> List:
    """A JSON list (datasets/default_charset.json layout) or a pickle of a list (data/HWDB_v1/charset_full.pkl layout)."""
    path = path or os.path.join(HERE, "data", "default_charset.json")
    if path.endswith(".json"):
        with open(path, encoding="utf-8") as f:
            return list(json.load(f))
    with open(path, "rb") as f:
        return list(pickle.load(f))


def load_labels(path: str, mode: str) -> List[Tuple[str, str]]:
    """-> [(image id / file name, text)] in dataset order.  Accepts the reference's labels.pkl ({"ground_truth": {split: [{"id",
    "text"}, ...]}}, datasets/IAM.py:57-60,77-80; mode "val" means "valid"), a JSON object {name: text} or list of [name, text],
    or a TSV file `name<TAB>text`."""
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f)             # the reference's own pickle of plain dicts/lists/strings
        split = "valid" if mode == "val" else mode
        return [(str(ex["id"]), ex["text"]) for ex in data["ground_truth"][split]]
    if path.endswith(".json"):
        with open(path, encoding="utf-8") as f:
            data = json.load(f)
        return [(str(k), v) for k, v in (data.items() if isinstance(data, dict) else data)]
    rows = []
    with open(path, encoding="utf-8") as f:
        for line in f:
            line = line.rstrip("\n")
            if line:
                name, _, text = line.partition("\t")
                rows.append((name, text))
    return rows


def find_image(folder: str, name: str) -> str:
    for cand in (name, name + ".jpg", name + ".png", name + ".jpeg"):
        p = os.path.join(folder, cand)
        if os.path.isfile(p):
            return p
    raise FileNotFoundError(f"{name}[.jpg|.png] not found under {folder}")


def read_rgb(path: str) -> np.ndarray:
    from PIL import Image                     # decoding the image FILE is host I/O; resize / normalise run on the device
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))  # datasets/IAM.py:87-89


def plan_batches(sizes: Sequence[Tuple[int, int]], batch: int, exact: bool, size: int, max_size: int) -> List[List[int]]:
    """Index batches.  exact: only lines with the same resized (h, w) share a batch (no padding); padded: neighbours in the
    width-sorted order share a batch."""
    resized = [get_size_with_aspect_ratio((w, h), size, max_size) for (h, w) in sizes]
    order = sorted(range(len(sizes)), key=lambda i: (resized[i][1], resized[i][0], i))
    batches: List[List[int]] = []
    for i in order:
        if batches and len(batches[-1]) < batch and (not exact or resized[batches[-1][0]] == resized[i]):
            batches[-1].append(i)
        else:
            batches.append([i])
    return batches


def image_size(path: str) -> Tuple[int, int]:
    """(h, w) from the file header (no pixel decode)."""
    from PIL import Image
    with Image.open(path) as im:
        w, h = im.size
    return int(h), int(w)


@torch.no_grad()
def predict_labels(model, images, batch: int = 32, exact: bool = True, TH: Optional[float] = None,
                   NM: Optional[float] = None, postprocessor=None, device="cuda", size: int = EVAL_SIZE,
                   max_size: int = EVAL_MAX_SIZE, rank: int = 0, world: int = 1, sizes: Optional[Sequence[Tuple[int, int]]] = None,
                   skip_errors: bool = True) -> List[Optional[List[int]]]:
    """convert_output_to_pred (evaluation.py:94-158) for a list of RGB uint8 images -> one label list per image, dataset
    order.  TH / NM given: the NMS decoder; otherwise the blank/argmax decoder with eps = 0.03 / C.
    `images`: a sequence of [h, w, 3] uint8 arrays, or (with `sizes` = the (h, w) of every line) a callable i -> array that is
    invoked only for the lines of this rank's shard.  skip_errors (the reference's behaviour, evaluation.py:498-504): a line whose
    load / forward / decode raises is reported and returned as None; KeyboardInterrupt always propagates."""
    lazy = callable(images)
    if lazy and sizes is None:
        raise ValueError("predict_labels: a loader callable needs `sizes`")
    sizes = list(sizes) if sizes is not None else [im.shape[:2] for im in images]
    n = len(sizes)
    load = images if lazy else (lambda i: images[i])
    batches = plan_batches(sizes, batch, exact, size, max_size)
    lo, hi = ddist.shard_bounds(len(batches), rank, world)
    tf = EvalTransform(size, max_size)
    nq = model.num_queries
    rec = torch.full((n, nq + 2), -1, dtype=torch.int32)          # labels[nq] | length | status (-1 not mine, 0 decoded, 1 skipped)

    def run(idx):
        samples = tf([load(i) for i in idx], device=device)
        out = model(samples)
        preds = E.decode_nms(out, postprocessor, TH, NM) if (TH is not None and NM is not None) else E.decode_blank(out)
        for i, p in zip(idx, preds):
            rec[i, : len(p)] = torch.tensor(p, dtype=torch.int32)
            rec[i, nq], rec[i, nq + 1] = len(p), 0

    for b in batches[lo:hi]:
        try:
            run(b)
        except KeyboardInterrupt:
            raise
        except Exception as e:
            if not skip_errors:
                raise
            for i in b:                                          # retry alone: only the offending line is lost
                try:
                    run([i])
                except KeyboardInterrupt:
                    raise
                except Exception as e1:
                    print(f"An error occurred affecting the metrics computation (line {i}: {type(e1).__name__}: {e1})", file=sys.stderr)
                    rec[i, :nq] = -1
                    rec[i, nq], rec[i, nq + 1] = 0, 1
            del e
    if world > 1:                               # every line is owned by exactly one rank: element-wise max merges the shards
        import torch.distributed as dist
        t = rec.to(device) if dist.get_backend() == "nccl" else rec
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rec = t.cpu()
    return [None if int(rec[i, nq + 1]) == 1 else rec[i, : int(rec[i, nq])].tolist() for i in range(n)]


def evaluate_predictions(pred_labels: Sequence[Sequence[int]], gt_texts: Sequence[str], charset: Sequence, dataset: str = "IAM",
                         metrics: str = "default", unicode_charset: bool = False) -> Dict:
    """The per-sample metric bookkeeping of evaluation.py:495-581 on already decoded predictions."""
    cs = list(charset)
    CER_list, WER_list, AR_list, CR_list, WA_list = [], [], [], [], []
    dict_char: Dict[int, int] = {}
    preds_str, gts_str, dists, lens = [], [], [], []
    conv = (lambda c: chr(c)) if unicode_charset else (lambda c: c)
    for pred, text in zip(pred_labels, gt_texts):
        if pred is None:                                                           # skipped line (evaluation.py:501-504: `continue`)
            continue
        gt = [cs.index(ord(c) if unicode_charset else c) for c in text]            # datasets/IAM.py:66-72
        if len(pred) > 0:                                                          # evaluation.py:340-352
            cer_it, dict_char, _ = E.character_error_rate_with_impact(list(pred), gt, dict_char)
        else:
            cer_it = 1
        preds_str.append("".join(conv(cs[int(i)]) for i in pred))
        gts_str.append("".join(conv(cs[int(i)]) for i in gt))
        wer_it = None
        if dataset in CER_DATASETS:                                                # :521-535 running string-level CER
            pg, pp = E.process_pred_string(gts_str[-1]), E.process_pred_string(preds_str[-1])
            dists.append(E.levenshtein(pg, pp))
            lens.append(len(pg))
            cer_it = sum(dists) / sum(lens)
        if metrics == "default":                                                   # :544-549 (argument order as the reference calls it)
            wer_it = E.word_error_rate(E.split_labels_into_words(gt, cs), E.split_labels_into_words(list(pred), cs))
            CER_list.append(cer_it)
            WER_list.append(wer_it)
        elif metrics == "CER_only":
            CER_list.append(cer_it)
        elif metrics == "chinese":                                                 # :560-565
            CER_list.append(cer_it)
            AR_list.append(1 - cer_it)
            CR_list.append(E.compute_cr(gt, list(pred)))
        elif metrics == "cipher":                                                  # :572-575
            CER_list.append(cer_it)
            WA_list.append(E.compute_wa(gt, list(pred)))
        else:
            raise ValueError(f"unknown --metrics {metrics}")

    def mean_ci(v):
        return (float(np.mean(v)), float(np.std(v) * 1.96 / np.sqrt(len(v)))) if len(v) else (float("nan"), float("nan"))

    return dict(CER_list=CER_list, WER_list=WER_list, AR_list=AR_list, CR_list=CR_list, WA_list=WA_list, dict_char=dict_char,
                list_preds_str=preds_str, list_gt_str=gts_str, cer=mean_ci(CER_list), wer=mean_ci(WER_list), ar=mean_ci(AR_list),
                cr=mean_ci(CR_list), wa=mean_ci(WA_list))


def write_outputs(res: Dict, out_dir: str, dataset: str, TH, NM) -> str:
    """The files evaluation.py:584-656 writes (except the PNG)."""
    stats_dir = os.path.join(out_dir, dataset)
    os.makedirs(stats_dir, exist_ok=True)
    np.save(os.path.join(stats_dir, "cer_list.npy"), res["CER_list"])
    with open(os.path.join(stats_dir, "dict_char.json"), "w") as f:
        json.dump(res["dict_char"], f)
    with open(os.path.join(stats_dir, "list_preds.txt"), "w", encoding="utf-8") as fp, \
            open(os.path.join(stats_dir, "list_gt.txt"), "w", encoding="utf-8") as fg:
        for p, g in zip(res["list_preds_str"], res["list_gt_str"]):
            fp.write(f"{p}\n")
            fg.write(f"{g}\n")
    with open(os.path.join(stats_dir, f"cer_TH_{TH}_NMS_{NM}.txt"), "w") as f:
        f.write(f"CER (TH={TH}) (NMS={NM}): {res['cer'][0]:.4f} +- {res['cer'][1]:.4f}")
    return stats_dir


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="python -m dtlr_amd.evaluation", description=__doc__.split("\n\n")[0])
    # the reference's flags (evaluation.py:15-27)
    ap.add_argument("--dataset", default="IAM")
    ap.add_argument("--mode", default="val")
    ap.add_argument("--new_class_embedding", action="store_true")
    ap.add_argument("--new_label_enc", action="store_true")
    ap.add_argument("--NMS_inference", action="store_true")
    ap.add_argument("--metrics", default="default", choices=["default", "CER_only", "chinese", "cipher"])
    ap.add_argument("--unicode", action="store_true")
    ap.add_argument("--weights", default="checkpoint.pth")
    ap.add_argument("--config", default="latin", help="a reference config file (config/*.py) or a preset: latin | chinese | tiny")
    ap.add_argument("--fix_enc_out_class", action="store_true")
    ap.add_argument("--TH", type=float, default=None)
    ap.add_argument("--NMS", type=float, default=None)
    # what the reference takes from its dataset registry (datasets/config.json + build_dataset)
    ap.add_argument("--images", required=True, help="folder with the line images (<id>.jpg / .png)")
    ap.add_argument("--labels", required=True, help="labels.pkl (reference layout), .json or .tsv")
    ap.add_argument("--charset", default=None, help="charset file (.json list / .pkl list); default datasets/default_charset.json")
    ap.add_argument("--out", default="stats_dect")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--batching", default="exact", choices=["exact", "padded"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32s", "f32"],
                    help="bf16 / f16: the 16-bit engines; f32s: fp32 activations, split fp16 products (parity-grade, ~1/3 of the 16-bit rate); f32: exact-fp32 MFMA")
    ap.add_argument("--limit", type=int, default=0, help="evaluate only the first N lines")
    ap.add_argument("--size", type=int, default=EVAL_SIZE, help="eval resize: short side (config/coco_transformer.py:1)")
    ap.add_argument("--max_size", type=int, default=EVAL_MAX_SIZE, help="eval resize: long-side cap (config/coco_transformer.py:2)")
    return ap


def main(argv: Optional[Sequence[str]] = None) -> Dict:
    args = build_parser().parse_args(argv)
    from .dino import DINO, PostProcess
    rank, local, world = ddist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("dtlr_amd.evaluation needs an MI355X (no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    charset = load_charset(args.charset)
    if args.config in ("latin", "chinese"):
        cfg = {"latin": DTLRConfig.latin, "chinese": DTLRConfig.chinese}[args.config]()
    elif args.config == "tiny":                 # reduced network of the test-suite (same topology, KB-sized assets)
        cfg = DTLRConfig.tiny(num_classes=len(charset))
    else:
        cfg = DTLRConfig.from_reference_file(args.config)
    rows = load_labels(args.labels, args.mode)
    if args.limit:
        rows = rows[: args.limit]
    model = DINO(cfg, compute_dtype={"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32s": "f32s"}[args.dtype])
    model = E.load_model(model, args.weights, device=dev, new_class_embedding=args.new_class_embedding, charset_size=len(charset),
                         new_label_enc=args.new_label_enc, fix_enc_out_class=args.fix_enc_out_class)
    # TH / NM grids exactly as evaluation.py:38-49
    if args.NMS is not None and args.TH is not None:
        list_TH, list_NM, nms_inference = [args.TH], [args.NMS], True
    elif not args.NMS_inference:
        list_TH, list_NM, nms_inference = [None], [None], False
    else:
        list_TH = list_NM = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]
        nms_inference = True
    paths = [find_image(args.images, name) for name, _ in rows]
    def header_size(p):                                        # an unreadable file is a per-sample error: it fails (and is skipped) when its shard decodes it
        try:
            return image_size(p)
        except Exception:
            return (size_fallback, size_fallback)
    size_fallback = 32
    sizes = [header_size(p) for p in paths]                    # headers only; pixels are decoded per shard, inside predict_labels
    images = lambda i: read_rgb(paths[i])                      # noqa: E731
    texts = [t for _, t in rows]
    post = PostProcess(num_select=cfg.num_select, nms_iou_threshold=cfg.nms_iou_threshold)
    last = {}
    for TH in list_TH:
        for NM in list_NM:
            preds = predict_labels(model, images, args.batch, args.batching == "exact", TH, NM, post, dev, args.size, args.max_size,
                                   rank=rank, world=world, sizes=sizes)
            if rank == 0:
                res = evaluate_predictions(preds, texts, charset, args.dataset, args.metrics, args.unicode)
                d = write_outputs(res, args.out, args.dataset, TH, NM)
                tail = f", TH {TH}, NM {NM}" if nms_inference else ""
                if args.metrics == "chinese":
                    print(f"AR {res['ar'][0]:.6f} +- {res['ar'][1]:.6f}, CR {res['cr'][0]:.6f} +- {res['cr'][1]:.6f}, lines {len(preds)}{tail}")
                elif args.metrics == "cipher":
                    print(f"SER {res['cer'][0]:.6f} +- {res['cer'][1]:.6f}, WA {res['wa'][0]:.6f} +- {res['wa'][1]:.6f}, lines {len(preds)}{tail}")
                elif args.metrics == "CER_only":
                    print(f"cer {res['cer'][0]:.6f} +- {res['cer'][1]:.6f}, lines {len(preds)}{tail}")
                else:
                    print(f"cer {res['cer'][0]:.6f} +- {res['cer'][1]:.6f}, wer {res['wer'][0]:.6f} +- {res['wer'][1]:.6f}, lines {len(preds)}{tail}")
                print(f"wrote {d}", file=sys.stderr)
                last = res
            if not nms_inference:
                break
        if not nms_inference:
            break
    ddist.finalize()
    return last


if __name__ == "__main__":
    main()
