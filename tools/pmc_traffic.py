#!/usr/bin/env python3
"""HBM traffic per launch from the rocprofv3 PMC counters, attributed to the engine's GEMM shapes.

    # on the GPU box, two separate passes (MI355X_MICROARCH.md: one counter per pass, --kernel-trace only):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -o pmc -- python tools/pmc_traffic.py workload $D/order.json
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -o pmc -- python tools/pmc_traffic.py workload $D/order.json
    python tools/pmc_traffic.py reduce $D/fetch $D/write $D/order.json profiles/r02_gemm_traffic.json

`workload` runs the bench configuration (Latin bf16, 32 lines of 128x2048): one untimed step, then one step with every MFMA-class
launch recorded IN ORDER (kind, tag, algorithmic bytes), then a 1 GiB elementwise pass that calibrates the counters.  `reduce` walks
the per-dispatch counter rows between the two marker dispatches in dispatch order, keeps the kernels of each family, and pairs them with the
recorded order -- so every `M.. N.. K..` shape gets its own measured bytes, not a per-symbol average over unlike shapes.

Corrections (the guide's HBM section): the counters are in KiB; gfx950 tallies a 128-byte request as 64 bytes, so the factor that
turns FETCH_SIZE into bytes is measured on the calibration pass (read 1 GiB, write 1 GiB) rather than assumed."""
import glob
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAMILIES = {
    "gemm": re.compile(r"dtlr::(gemm_ws_kernel|gemm_ws_tall_kernel|gemm_k256_kernel|gemm_kres_kernel|conv3x3_patch_kernel|gemm_nt_kernel|dec_query_stage_kernel)\b"),
    "proj_ln": re.compile(r"dtlr::proj_ln_\w*kernel\b"),
    "ffn": re.compile(r"dtlr::(ffn3_bf16_kernel<|ffn2_bf16_kernel<|ffn_fused_bf16_kernel<0, false>)"),
    "msda_enc": re.compile(r"dtlr::msda_enc_lds_kernel\b"),
}
KIND_FAMILY = {"gemm_bf16": "gemm", "gemm_f32": "gemm", "proj_ln_bf16": "proj_ln", "ffn_fused_bf16": "ffn"}
CALIB_ELEMS = 1 << 28           # fp32: 1 GiB read + 1 GiB written


def workload(order_path):
    import torch
    from dtlr_amd import ops, synth, weights
    from dtlr_amd.config import DTLRConfig
    from dtlr_amd.engine import DTLREngine
    dev = torch.device("cuda", 0)
    cfg = DTLRConfig.latin()
    eng = DTLREngine(cfg, weights.synthetic_state_dict(cfg, seed=0), dev, torch.bfloat16)
    lines = synth.noise_lines(32, 128, [2048] * 32, seed=1000)
    x = torch.stack([l if torch.is_tensor(l) else torch.from_numpy(l) for l in lines]).float().to(dev)
    mask = torch.zeros((32, 128, 2048), dtype=torch.bool, device=dev)
    eng.forward(x, mask, has_padding=False)
    torch.cuda.synchronize()
    ops.MFMA_EVENTS = []
    ops.MFMA_EVENTS_MIN_FLOPS = 0.0
    marker = torch.ones(64, dtype=torch.float32, device=dev)
    torch.lgamma(marker)                     # a kernel the engine never launches: brackets the recorded step in the dispatch stream
    eng.forward(x, mask, has_padding=False)
    torch.lgamma(marker)
    torch.cuda.synchronize()
    order = [{"kind": e[2], "flops": e[3], "bytes": e[4], "tag": e[5]} for e in ops.MFMA_EVENTS]
    ops.MFMA_EVENTS = None
    src = torch.ones(CALIB_ELEMS, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    for _ in range(3):
        torch.neg(src, out=dst)
    torch.cuda.synchronize()
    os.makedirs(os.path.dirname(os.path.abspath(order_path)), exist_ok=True)
    with open(order_path, "w") as f:
        json.dump(order, f)
    print(f"pmc_traffic workload: {len(order)} MFMA-class launches in the recorded step", flush=True)


def dispatches(d, counter):
    """[(kernel_name, value)] in dispatch order, one entry per dispatch (instances summed)."""
    dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    if not dbs:
        raise SystemExit(f"no rocpd database under {d}")
    con = sqlite3.connect(dbs[0])
    rows = con.execute("select dispatch_id, kernel_name, sum(value) from counters_collection where counter_name = ? "
                       "group by dispatch_id, kernel_name order by dispatch_id", (counter,)).fetchall()
    con.close()
    rows = [(r[1], float(r[2])) for r in rows]
    marks = [i for i, (n, _) in enumerate(rows) if "lgamma" in n]
    if len(marks) != 2:
        raise SystemExit(f"{counter}: expected 2 marker dispatches, found {len(marks)}")
    return rows[marks[0] + 1:marks[1]], rows


def reduce_(fetch_dir, write_dir, order_path, out_path):
    order = json.load(open(order_path))
    fetch, fetch_all = dispatches(fetch_dir, "FETCH_SIZE")
    write, write_all = dispatches(write_dir, "WRITE_SIZE")

    def calib(rows):
        v = [val for (name, val) in rows if "neg" in name and val > 1e5]
        if not v:
            raise SystemExit("calibration dispatches not found")
        return sum(v) / len(v)

    gib_kib = CALIB_ELEMS * 4 / 1024.0
    f_factor = gib_kib / calib(fetch_all)          # bytes really read per counted KiB / 1024
    w_factor = gib_kib / calib(write_all)
    out = {"_source": "tools/pmc_traffic.py: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over one engine step "
                      "(Latin bf16, 32 x 128x2048); per-dispatch rows paired in dispatch order with the launch order ops._Timed recorded",
           "_calibration": {"pass": "torch.neg over 2^28 fp32 (1 GiB read, 1 GiB written) in the same process",
                            "fetch_bytes_per_counted_byte": round(f_factor, 4), "write_bytes_per_counted_byte": round(w_factor, 4)},
           "_units": "bytes per launch = (FETCH_SIZE KiB x fetch factor + WRITE_SIZE KiB x write factor) x 1024"}
    per_family = {}
    for fam, rx in FAMILIES.items():
        f = [v for (n, v) in fetch if rx.search(n)]
        w = [v for (n, v) in write if rx.search(n)]
        per_family[fam] = (f, w)
    # msda encoder: every dispatch is the same shape
    f, w = per_family["msda_enc"]
    if f and w:
        fb, wb = 1024.0 * f_factor * sum(f) / len(f), 1024.0 * w_factor * sum(w) / len(w)
        out["bf16_enc_kernel"] = "msda_enc_lds_kernel (engine configuration: bf16 value, bf16 projection row), B=32, S=Lq=5440"
        out["bf16_enc_fetch_bytes"] = round(fb)
        out["bf16_enc_write_bytes"] = round(wb)
        out["bf16_enc_bytes_per_launch"] = round(fb + wb)
    shapes, classes = {}, {}
    for fam in ("gemm", "proj_ln", "ffn"):
        want = [o for o in order if KIND_FAMILY.get(o["kind"]) == fam]
        f, w = per_family[fam]
        if not want:
            continue
        if len(f) != len(w) or len(f) < len(want):
            out[f"_error_{fam}"] = f"{len(f)} fetch / {len(w)} write dispatches for {len(want)} recorded launches"
            continue
        if len(f) != len(want):
            # a recorded launch is several dispatches (the encoder FFN = a 192-row pass + a 128-row remainder pass): class mean only
            b = 1024.0 * (f_factor * sum(f) + w_factor * sum(w))
            kind = want[0]["kind"]
            out[f"{kind}_bytes_per_launch_mean"] = round(b / len(want))
            out[f"{kind}_algorithmic_bytes_per_launch_mean"] = round(sum(o["bytes"] for o in want) / len(want))
            out[f"{kind}_launches"] = len(want)
            out[f"{kind}_dispatches"] = len(f)
            continue
        for o, fv, wv in zip(want, f, w):
            b = 1024.0 * (f_factor * fv + w_factor * wv)
            if o["tag"] and fam == "gemm":
                s = shapes.setdefault(o["tag"], [0.0, 0, 0.0])
                s[0] += b; s[1] += 1; s[2] += o["bytes"]
            if o["flops"] >= 2.0e9:                       # the set bench.py times
                c = classes.setdefault(o["kind"], [0.0, 0, 0.0])
                c[0] += b; c[1] += 1; c[2] += o["bytes"]
    for tag, (b, n, alg) in sorted(shapes.items()):
        out[f"gemm:{tag}"] = round(b / n)
        out[f"gemm_alg:{tag}"] = round(alg / n)
    for kind, (b, n, alg) in classes.items():
        out[f"{kind}_bytes_per_launch_mean"] = round(b / n)
        out[f"{kind}_algorithmic_bytes_per_launch_mean"] = round(alg / n)
        out[f"{kind}_launches"] = n
    # per kernel SYMBOL (round 5: bench.py's `roofline` is one kernel, named as rocprofv3 names it): mean bytes per dispatch of every dtlr kernel
    # of the recorded step.  Symbols whose dispatches have unlike shapes (the tiled GEMM's instantiations) are means over those shapes.
    sym_f, sym_w = {}, {}
    for (rows, acc) in ((fetch, sym_f), (write, sym_w)):
        for n, v in rows:
            m = re.search(r"dtlr::([A-Za-z0-9_]+(?:<[^(]*>)?)\(", n)
            if m:
                a = acc.setdefault(m.group(1), [0.0, 0])
                a[0] += v; a[1] += 1
    for sym in sorted(sym_f):
        if sym in sym_w and sym_f[sym][1] == sym_w[sym][1]:
            nd = sym_f[sym][1]
            out[f"symbol:{sym}"] = {"bytes_per_dispatch_mean": round(1024.0 * (f_factor * sym_f[sym][0] + w_factor * sym_w[sym][0]) / nd),
                                    "fetch": round(1024.0 * f_factor * sym_f[sym][0] / nd), "write": round(1024.0 * w_factor * sym_w[sym][0] / nd), "dispatches": nd}
    out["_commit"] = os.environ.get("DTLR_COMMIT") or "unknown (set DTLR_COMMIT=$(git rev-parse --short HEAD) when launching through gpurun: .git does not travel)"
    with open(out_path, "w") as fp:
        json.dump(out, fp, indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("gemm") and not k.startswith("symbol:")}, indent=1))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "workload":
        workload(sys.argv[2])
    elif len(sys.argv) == 6 and sys.argv[1] == "reduce":
        reduce_(*sys.argv[2:6])
    else:
        raise SystemExit(__doc__)
