#!/usr/bin/env python3
"""bench.py -- DTLR inference hot path on MI355X: text-lines/sec on synthetic 128x2048 crops.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --config chinese            # BASELINE.json configs[4]: 7356-class head, mixed-length 128x2560, padded

A "step" = one pass of the hot path over one batch of 32 synthetic lines per GPU, inputs already resident in HBM: DINO forward
(backbone -> encoder -> two-stage -> decoder -> heads) + the blank decoder producing per-line label records (+ ONE RCCL
all-gather of those records when N > 1).  Workload = BASELINE.json configs[1]: Latin model (C=166), bf16, bs=32, 128x2048,
random-init name-seeded weights with trained-like head margins (generator v2; no checkpoint ships with the reference).
Rank 0 prints ONE JSON line on stdout, kept below 6 KB (compact_line: the contract's keys, `roofline` of the dominant kernel, `roofline_msda`,
`cpu_baseline`, `distributed`, six numbers per engine in `by_dtype`); the FULL result -- everything described below -- is written to
gpurun_out/bench_detail.json (or ./bench_detail.json) and to stderr.

timing       : W warm-up steps, then blocks of exactly K steps, each bracketed by barrier + torch.cuda.synchronize() on both
               sides and reduced with MAX over ranks.  The block is repeated until >= 2 s have been timed (a 20-step block is
               0.2 s: one clock ramp away from noise) and the MEDIAN block is reported: ms_per_step = median block / K,
               value = global lines per block / median block time.  `block_ms` lists every block.
data-parallel: the global batch is ONE seeded list of n_gpus x 32 lines; rank r owns the contiguous shard shard_bounds(r)
               (dtlr_amd/dist.py).  After the timed region rank 0 recomputes every shard itself and compares the all-gathered
               records bit for bit (`dp_verified`): N-rank output == single-process output on the same lines, in order.
parity       : `parity_vs_oracle` (oracle/parity.py): the CPU oracle runs --parity-lines (16) lines of the benched batch ONCE, with its
               own two-stage selection; every engine is compared against that run twice: `free_running` = the engine's decoded strings
               (its own selection) against the oracle's, as they come out -- `strings_identical_free_running: k/n`, no tolerance, no
               accounting -- and `teacher_forced` = the oracle's decoder re-run on the engine's selection: max logit / box / cx error
               against a FIXED per-engine budget (fp32-grade engines: north_star's 1e-3) and the strings on the same selection.
               `free_running_v4`: the free-running leg on generator-v4 weights (dtlr_amd/weights.py: identical content queries, characters
               read from the image -- rank-invariant like a trained recogniser; the benched v2 weights plant a character per selection
               RANK, so on them the leg measures rank swaps): the engine rebuilt on v4 weights, 8 lines, strings as they come out.
               `by_dtype`: the same batch through the other engines (bf16 = libdtlr_hip.so, f16 = libdtlr_hip_f16.so, f32s = split
               fp32, f32 = exact fp32): a short timed block (lines/s) and the same legs each.
roofline     : per KERNEL, measured live with HIP events on the launch stream: MSDA inside the timed blocks, the MFMA kernels (fused FFN,
               GEMM, projection+norm) in a replay of the same steps right after them (an event pair per launch inside the timed region
               would cost ~2.5 ms of stream time per step).  `roofline` = the single kernel with the largest share of the step
               (`symbol` = its name in profiles/*_kernel_stats.csv, so achieved = algorithmic flops per launch / average launch time
               can be recomputed from that file); `roofline_by_kernel` rows carry `symbol` when they are one kernel and
               `class: true` when they aggregate several (the tiled GEMM's instantiations: listed per shape in `gemm_by_shape`,
               each with its MFMA fraction AND the HBM fraction of its compulsory bytes).
               `traffic` = PMC-measured HBM bytes per launch (profiles/*_traffic.json: rocprofv3 --pmc FETCH_SIZE /
               WRITE_SIZE passes, gfx950 correction per MI355X_MICROARCH.md) when a measurement for that kernel is committed.
cpu_baseline : SURVEY.md 8(d): the CPU oracle (oracle/dtlr_oracle.py, a port pinned to the reference through tests/golden;
               MSDA through the reference's own grid_sample formulation) on BASELINE configs[0] -- 16 synthetic 128x2048 lines,
               fp32, 1 warm-up + 3 timed forwards+decodes, lines/s = 16 / median -- at 8 threads and at all physical cores
               (time-capped), on rank 0 at N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

T_START = time.perf_counter()


def log(msg: str) -> None:
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak (no sparsity); fp32 MFMA (16x16x4_f32) class: 157
MFMA_PEAK_F32_TFLOPS = 157.0
MIN_TIMED_SECONDS = 2.0


def msda_algorithmic_bytes_per_line(S: int, Lq: int, value_elem: int, M=8, D=32, L=4, P=4, ref_dim=2) -> int:
    """Compulsory bytes of one MSDA call for one line ("all operands in the element size e"): read
    value once, read the offsets + attention logits of every (head, level, point) once (3 numbers each: the fused kernel
    takes them as the raw projection row, in the engine dtype), read the fp32 reference points, write out."""
    return S * M * D * value_elem + Lq * M * L * P * 3 * value_elem + Lq * L * ref_dim * 4 + Lq * M * D * value_elem


def physical_cores() -> int:
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    try:
        pairs = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
                pairs.add((phys, core))
        if pairs:
            return len(pairs)
    except Exception:
        pass
    return os.cpu_count() or 8


def cpu_baseline(n_lines: int = 16, height: int = 128, width: int = 2048, timed: int = 3, cap_s: float = 75.0):
    """SURVEY.md 8(d) / BASELINE.md: >= 1 warm-up + >= 3 timed batches of 16 lines, lines/s = 16 / median, at N = 8 threads (the
    survey probe's setting: 0.92 lines/s with the reference itself) and at N = all physical cores.  Each leg is time-capped:
    torch's CPU ops regress badly far beyond ~32 threads on many-core hosts (0.11 lines/s at 128 threads was measured in round
    1), so a leg whose warm-up already exceeds the cap reports that single run."""
    from dtlr_amd import synth, weights
    from dtlr_amd.config import DTLRConfig
    from oracle import dtlr_oracle as O      # the reported CPU baseline (a port); never the product path
    O.MSDA_CORE = "grid_sample"              # the reference's own CPU formulation (ms_deform_attn_func.py:41-61)
    cfg = DTLRConfig.latin()
    sd = weights.synthetic_state_dict(cfg, seed=0)
    imgs = synth.noise_lines(n_lines, height, width, seed=123)
    nphys = physical_cores()

    def run():
        t0 = time.perf_counter()
        out = O.dino_forward(sd, cfg, imgs)
        O.decode_blank(out)
        return time.perf_counter() - t0

    legs = {}
    for th in sorted({min(8, nphys), nphys}):
        torch.set_num_threads(th)
        warm = run()
        ts = []
        spent = warm
        leg_timed = timed if th <= 8 else (timed if warm < 10.0 else 0)      # the all-cores leg is only repeated when it is competitive: torch's
        while len(ts) < leg_timed and spent < cap_s:                          # CPU ops regress on many-core hosts (0.37 lines/s at 128 threads)
            ts.append(run())
            spent += ts[-1]
        used = ts if ts else [warm]
        med = sorted(used)[len(used) // 2]
        legs[th] = {"lines_per_s": round(n_lines / med, 4), "median_s": round(med, 3), "timed_runs": len(ts), "warmup_s": round(warm, 3)}
        log(f"cpu_baseline: {th} threads: warm-up {warm:.2f}s, timed {['%.2f' % t for t in ts]} -> {n_lines / med:.3f} lines/s")
    O.MSDA_CORE = "gather"
    best = max(legs, key=lambda k: legs[k]["lines_per_s"])
    return {"value": legs[best]["lines_per_s"], "unit": "lines/s", "cores": best, "kind": "port",
            "physical_cores": nphys, "by_threads": {str(k): v for k, v in legs.items()},
            "sample_short": f"configs[0]: {n_lines} synthetic {height}x{width} fp32 lines, oracle forward + blank decode; `value` = the faster of the 8-thread and "
                            f"all-{nphys}-core legs (a leg with timed_runs 0 is a single run: its warm-up exceeded the time cap)",
            "sample": f"BASELINE configs[0]: {n_lines} synthetic {height}x{width} fp32 lines, oracle forward + blank decode (MSDA = the reference's "
                      f"grid_sample core), 1 warm-up + up to {timed} timed runs per leg (median; {cap_s:.0f} s cap per leg), legs at 8 threads and at "
                      f"all {nphys} physical cores; `value` is the faster leg"}


def msda_offset_sensitivity(eng, B, dtype, level_hw, sigmas=(0.0, 8.0, 32.0), iters=10):
    """The encoder's deformable-sampling call as a function of how far a checkpoint's offset heads look (the synthetic weights plant
    <= 4 px; a trained line recogniser may look further along the line): offsets N(0, sigma^2) pixels of the sampled level, B lines,
    the bench's level shapes.  Per sigma: the kernel the engine's calibration rule picks (LDS windows at a halo of 8 / 16 / 24 columns,
    or the gather kernel), its time, and the two fixed choices beside it."""
    import math
    from dtlr_amd import ops
    dev = eng.device
    M, L, P = 8, 4, 4
    S = sum(h * w for h, w in level_hw)
    shapes = torch.as_tensor(level_hw, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    g = torch.Generator(device="cpu").manual_seed(0)
    value = (torch.rand((B, S, M, 32), generator=g) * 2 - 1).to(dev).to(dtype)
    rp = torch.cat([torch.stack(torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h, torch.linspace(0.5, w - 0.5, w) / w, indexing="ij")[::-1], -1).reshape(-1, 2)
                    for h, w in level_hw], 0)
    refg = rp[None, :, None, :].expand(B, S, L, 2).contiguous().to(dev)
    alg = msda_algorithmic_bytes_per_line(S, S, 2) * B

    def timeit(fn):
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    rows = []
    for sigma in sigmas:
        ow = torch.randn((B, S, M * L * P * 3), generator=g).to(dev)
        ow[..., : M * L * P * 2] *= sigma
        ow = ow.to(dtype)
        best = ("gather", None, 1.0)
        far = {}
        for halo, base in eng.msda_halo_base.items():
            if ops.msda_encoder_fits(level_hw, dtype, halo):
                far[halo] = ops.msda_encoder_far_fraction(dtype, level_hw, ow, refg, M, halo)
                c = base + eng.msda_far_slope * math.sqrt(far[halo])
                if c < best[2]:
                    best = ("lds", halo, c)
        t_lds8 = timeit(lambda: ops.msda_encoder(value, level_hw, ow, refg, 8))
        t_gather = timeit(lambda: ops.msda_fused(value, shapes, lsi, ow, refg))
        t_pick = t_gather if best[0] == "gather" else (t_lds8 if best[1] == 8 else timeit(lambda: ops.msda_encoder(value, level_hw, ow, refg, best[1])))
        rows.append({"offset_sigma_px": sigma, "far_fraction_halo8": round(far.get(8, float("nan")), 5), "picked": best[0] + (f"@halo{best[1]}" if best[1] else ""),
                     "picked_ms": round(t_pick, 4), "picked_gbps_algorithmic": round(alg / t_pick / 1e6, 1), "frac_of_hbm_peak": round(alg / t_pick / 1e6 / HBM_PEAK_GBS, 4),
                     "lds_halo8_ms": round(t_lds8, 4), "gather_ms": round(t_gather, 4)})
    return rows


def _sci(v: float) -> float:
    """4 significant digits, whatever the magnitude (round(3e-5, 4) used to print the fp32 engine's error as 0.0)"""
    return float(f"{v:.3e}")


STDOUT_LINE_LIMIT = 6000        # bytes: the driver parses ONE stdout line; round 5's 22 KB line came back `parsed: null`


def _tf(p):
    return (p or {}).get("teacher_forced") or {}


def _fr(p):
    return (p or {}).get("free_running") or {}


def _compact_roof(r):
    """the dominant kernel's roofline object: SURVEY 8(d)'s fields + what lets a reader recompute `achieved` from profiles/*_kernel_stats.csv"""
    if not r:
        return None
    keep = ("bound", "symbol", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch",
            "mean_launch_ms", "launches_timed", "ms_per_step")
    out = {k: r[k] for k in keep if r.get(k) is not None}
    out.setdefault("traffic", None)
    if "symbol" not in out:
        out["kernel"] = str(r.get("kernel", ""))[:96]
    return out


def _compact_engine(ent):
    """six numbers per engine: rate, step time, teacher-forced logit error + strings on the benched weights, free-running strings + CER on v4"""
    if not isinstance(ent, dict) or "error" in ent:
        return {"error": str((ent or {}).get("error"))[:120]}
    pv, v4 = ent.get("parity_vs_oracle"), ent.get("free_running_v4")
    out = {"lines_per_s": ent.get("lines_per_s"), "ms_per_step": ent.get("ms_per_step")}
    if isinstance(pv, dict) and "error" not in pv:
        out.update(logit_err_max=_tf(pv).get("logit_err_max"), logit_budget=_tf(pv).get("logit_budget"),
                   strings_teacher_forced=_tf(pv).get("strings_identical_same_selection"), parity_gate=pv.get("parity_gate"))
    if isinstance(v4, dict) and "error" not in v4:
        out.update(strings_free_running_v4=_fr(v4).get("strings_identical_free_running"), cer_free_running_v4=_fr(v4).get("cer_free_running"))
    return out


def _all_of(frac) -> bool:
    """'16/16' -> True, '8/16' / None -> False"""
    try:
        a, b = str(frac).split("/")
        return int(b) > 0 and int(a) == int(b)
    except (ValueError, AttributeError):
        return False


def compact_line(full: dict) -> dict:
    """The ONE stdout line (< STDOUT_LINE_LIMIT bytes) from the full result: the contract's keys, the dominant kernel's `roofline`, the
    deformable-sampling kernel's HBM figure beside it (`roofline_msda`), `cpu_baseline`, `distributed` and six numbers per engine.  Everything
    else (roofline_by_kernel, gemm_by_shape, the full parity objects, block times) goes to bench_detail.json and stderr."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: full.get(k) for k in keys}
    cfg = full.get("config") or {}
    line["config"] = {k: cfg.get(k) for k in ("workload", "global_batch", "parallelism", "canvas", "num_classes", "backbone") if k in cfg}
    if cfg.get("engine_opts"):
        line["config"]["engine_opts"] = cfg["engine_opts"]
    line["timed_blocks"] = full.get("timed_blocks")
    d = full.get("distributed") or {}
    line["distributed"] = {k: d.get(k) for k in ("backend", "world_size", "rccl_ranks", "devices_visible", "dp_verified")}
    line["roofline"] = _compact_roof(full.get("roofline"))
    msda = next((r for r in full.get("roofline_by_kernel") or [] if str(r.get("symbol", "")).startswith("msda_enc")), None)
    if msda is not None and msda is not full.get("roofline"):
        line["roofline_msda"] = _compact_roof(msda)
    ts = full.get("traffic_source") or {}
    line["traffic_source"] = ts.get("file") if ts.get("attached") else None
    cb = full.get("cpu_baseline")
    if cb:
        c = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "physical_cores")}
        c["sample"] = cb.get("sample_short") or str(cb.get("sample", ""))[:200]
        legs = {}
        for th, leg in (cb.get("by_threads") or {}).items():
            runs = leg.get("timed_runs", 0)
            legs[th] = {"lines_per_s": leg.get("lines_per_s"), "runs": (f"median of {runs} timed" if runs else "single run (the warm-up; time cap)")}
        c["by_threads"] = legs
        line["cpu_baseline"] = c
        line["speedup_vs_cpu"] = full.get("speedup_vs_cpu")
    if "latency_ms_bs1" in full:
        line["latency_ms_bs1"] = full["latency_ms_bs1"]
    if full.get("latency_ms_bs1_graph") is not None:
        line["latency_ms_bs1_graph"] = full["latency_ms_bs1_graph"]
    if full.get("by_dtype"):
        line["by_dtype"] = {k: _compact_engine(v) for k, v in full["by_dtype"].items()}
        # the rate of the fastest engine that meets north_star's parity statement (logits within 1e-3 of the fp32 CPU oracle, identical strings
        # on the same selection AND free-running): `value` is the benchmark's bf16 configuration, which does NOT reproduce the strings
        grade = [(v.get("lines_per_s") or 0.0, k) for k, v in line["by_dtype"].items()
                 if v.get("logit_budget") is not None and v.get("logit_budget") <= 1e-3 and (v.get("logit_err_max") or 1.0) <= 1e-3
                 and _all_of(v.get("strings_teacher_forced")) and _all_of(v.get("strings_free_running_v4"))]
        if grade:
            best = max(grade)
            line["parity_grade"] = {"dtype": best[1], "lines_per_s": best[0]}
    if "observed_on_user_assets" in full:
        line["observed_on_user_assets"] = {k: v for k, v in full["observed_on_user_assets"].items() if k != "msda_encoder_choice_by_layer"}
    line["detail"] = full.get("detail_file")
    # belt and braces: whatever a future field adds, the stdout line stays parseable -- drop optional objects, largest first
    for k in ("observed_on_user_assets", "roofline_msda", "by_dtype", "timed_blocks", "traffic_source"):
        if len(json.dumps(line)) < STDOUT_LINE_LIMIT:
            break
        line.pop(k, None)
    return line


def write_detail(full: dict, path: str = "") -> str:
    """the full result beside the line: `path` (--detail) if given, else gpurun_out/bench_detail.json when that directory exists (it is merged
    back from the GPU box), else ./bench_detail.json; returns the path relative to the repo root (or "" when nothing could be written)"""
    if path:
        try:
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            return os.path.relpath(path, ROOT)
        except OSError:
            pass
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        if os.path.isdir(d) and os.access(d, os.W_OK):
            path = os.path.join(d, "bench_detail.json")
            try:
                with open(path, "w") as f:
                    json.dump(full, f, indent=1)
                return os.path.relpath(path, ROOT)
            except OSError:
                continue
    return ""


V4_PARITY_LINES = 8


def free_running_v4(cfg, sd4, ob4, engine_name, dtype, dev, x, mask, rows4, padded):
    """The free-running leg proper: the engine `engine_name` rebuilt on generator-v4 weights (dtlr_amd/weights.py: identical content queries,
    characters read from the image -- rank-invariant like a trained recogniser) on lines `rows4` of the benched batch, against the oracle's own
    run of the same weights and lines (ob4).  Strings as they come out: `strings_identical_free_running` k/n and `cer_free_running`."""
    from dtlr_amd.engine import DTLREngine
    from dtlr_amd.evaluation import decode_blank_records      # noqa: F401  (the product decoder runs inside compare() through the oracle's restatement of it on the engine's outputs)
    e4 = DTLREngine(cfg, sd4, dev, dtype, split=engine_name == "f32s")
    out = e4.forward(x[rows4].contiguous(), mask[rows4].contiguous(), has_padding=padded, return_debug=True)
    torch.cuda.synchronize()
    d = out["_debug"]
    all_rows = list(range(len(rows4)))
    r = ob4.compare(engine_name, out["pred_logits"][all_rows], out["pred_boxes"][all_rows], d["topk_idx"][all_rows], d["topk_scores"][all_rows], budgeted=False)
    r["rows"] = list(rows4)
    r["weights"] = "generator v4 (rank-invariant content queries, image-driven characters)"
    del e4, out
    torch.cuda.empty_cache()
    return r


def parity_vs_oracle(ob, engine_name, out, rows, chinese=False):
    """One engine's FREE-RUNNING outputs for lines `rows` of the benched batch against the oracle's run of the same lines
    (oracle/parity.py::OracleBatch.compare): `free_running` (strings as they come out) and `teacher_forced` (arithmetic error on the
    engine's own selection against the engine's fixed budget)."""
    d = out["_debug"]
    return ob.compare(engine_name, out["pred_logits"][rows], out["pred_boxes"][rows], d["topk_idx"][rows], d["topk_scores"][rows], chinese=chinese)


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher in the environment: start the N ranks ourselves, exactly as the documented
    command does (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...);
    rank 0 prints the one JSON line on the inherited stdout, a failing rank's exit code is ours."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {args.gpus} without a launcher: " + " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="lines per GPU")
    ap.add_argument("--config", default="latin", choices=["latin", "chinese", "latin-mixed", "latin-eval"],
                    help="latin = BASELINE configs[1] (the metric); chinese = configs[4]; latin-mixed = Latin model on seeded mixed widths "
                         "{1280..2048} zero-padded to 128x2048 with masks (what a real dataset looks like); latin-eval = uint8 128x2048 lines "
                         "through the reference's eval transform (short side 800 capped at 1333: 83x1328 canvases, datasets/transforms.py:78-142)")
    ap.add_argument("--backbone", default=None, help="override the config's backbone (e.g. swin_T_224_1k, swin_B_224_22k: models/dino/backbone.py:172-205)")
    ap.add_argument("--engine-opt", action="append", default=[], metavar="NAME=VALUE",
                    help="set a DTLREngine attribute for an A/B run (e.g. sort_queries=0, use_kres=0); recorded in config.engine_opts")
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=0, help="0 = 2048 (latin) / 2560 canvas with mixed widths (chinese)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32s", "f32"],
                    help="the timed engine: bf16 (BASELINE configs[1]), f16 (the fp16 build of the same kernels: same rate, 8x finer rounding), "
                         "f32s (fp32 activations, split fp16 products: parity-grade), f32 (exact-fp32 MFMA parity engine)")
    ap.add_argument("--no-other-dtypes", action="store_true", help="skip the short lines/s + parity legs of the two engines that are not --dtype")
    ap.add_argument("--parity-lines", type=int, default=16, help="lines of the benched batch the CPU oracle runs (parity_vs_oracle)")
    ap.add_argument("--weights", default=None, help="a reference checkpoint (checkpoint.pth: {'model': state_dict}) instead of the name-seeded synthetic weights "
                                                    "(BASELINE configs[2]); class count and backbone are taken from the tensors")
    ap.add_argument("--images", default=None, help="a folder of line images (png / jpg): the first --batch x N of them (sorted by name) through the "
                                                   "reference's eval transform replace the synthetic lines")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity_vs_oracle legs (an oracle forward of --parity-lines lines on the host)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); gloo for the single-GPU DP test")
    ap.add_argument("--single-device", action="store_true", help="all ranks use cuda:0 (tests: two ranks on one GPU need gloo)")
    ap.add_argument("--min-seconds", type=float, default=MIN_TIMED_SECONDS)
    ap.add_argument("--no-bs1", action="store_true", help="skip the single-line latency leg (latency_ms_bs1)")
    ap.add_argument("--detail", default="", help="where the FULL result (roofline_by_kernel, gemm_by_shape, parity objects ...) is written as JSON; "
                                                 "default gpurun_out/bench_detail.json, else ./bench_detail.json.  stdout carries the compact line only")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    import faulthandler
    faulthandler.dump_traceback_later(900, exit=False, file=sys.stderr)      # diagnose hangs on the box

    from dtlr_amd import dist as ddist
    from dtlr_amd import ops, synth, weights
    from dtlr_amd.config import DTLRConfig
    from dtlr_amd.engine import DTLREngine
    from dtlr_amd.evaluation import decode_blank_records
    import torch.distributed as tdist

    rank, local, world = ddist.init_from_env(args.backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    placement = ddist.pin_to_local_cpus(local, world) if (world > 1 and not args.single_device) else None
    DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32s": torch.float32}
    dtype = DT[args.dtype]

    chinese = args.config == "chinese"
    mixed, evalshape = args.config == "latin-mixed", args.config == "latin-eval"
    cfg = DTLRConfig.chinese() if chinese else DTLRConfig.latin()
    if args.backbone:
        import dataclasses
        cfg = dataclasses.replace(cfg, backbone=args.backbone)
    if args.weights:
        # BASELINE configs[2]: a user-supplied checkpoint (none ships with the reference: README.md:63-71).  The class count comes from the tensors
        import dataclasses
        sd = weights.load_checkpoint_state_dict(args.weights)
        cfg = dataclasses.replace(cfg, num_classes=weights.num_classes_of(sd))
        log(f"--weights {args.weights}: {len(sd)} tensors, {cfg.num_classes} classes")
    else:
        sd = weights.synthetic_state_dict(cfg, seed=0)
    eng = DTLREngine(cfg, sd, dev, dtype, split=args.dtype == "f32s")
    for kv in args.engine_opt:
        k, _, v = kv.partition("=")
        if not hasattr(eng, k):
            raise SystemExit(f"--engine-opt: DTLREngine has no attribute {k!r}")
        setattr(eng, k, type(getattr(eng, k))(int(v)) if isinstance(getattr(eng, k), (bool, int)) else float(v))
    log(f"engine packed ({args.dtype}, {args.config}), rank {rank}/{world}")
    B = args.batch
    n_total = B * world
    lo, hi = ddist.shard_bounds(n_total, rank, world)
    # ONE global seeded batch; a rank materialises only its shard (line i is seeded by i, whatever the shard)
    if chinese:
        canvas_w = args.width or 2560
        widths = synth.mixed_widths(n_total, [canvas_w - 1024, canvas_w - 768, canvas_w - 512, canvas_w - 256, canvas_w], seed=7)
        widths[::B] = [canvas_w] * len(widths[::B])                    # every shard has a full-width line: same canvas on every rank
    elif mixed:
        canvas_w = args.width or 2048
        widths = synth.mixed_widths(n_total, [canvas_w - 768, canvas_w - 512, canvas_w - 256, canvas_w], seed=11)
        widths[::B] = [canvas_w] * len(widths[::B])
    else:
        canvas_w = args.width or 2048
        widths = [canvas_w] * n_total

    def make_lines(a, b):
        return synth.noise_lines(b - a, args.height, widths[a:b], seed=1000, start=a)

    def to_batch(lines):
        x = torch.zeros((len(lines), 3, args.height, canvas_w), dtype=torch.float32)
        m = torch.ones((len(lines), args.height, canvas_w), dtype=torch.bool)
        for i, im in enumerate(lines):
            x[i, :, :, : im.shape[2]] = im
            m[i, :, : im.shape[2]] = False
        return x.to(dev), m.to(dev)

    preproc_ms = None
    user_images = None
    if args.images:
        # BASELINE configs[2]: real line crops through the reference's eval transform (datasets/transforms.py:78-142) on the device, padded to
        # one canvas with masks (util/misc.py:375-397); rank r takes files [lo, hi) of the sorted folder
        from dtlr_amd.eval_harness import read_rgb
        from dtlr_amd.transforms import preprocess_lines
        names = sorted(f for f in os.listdir(args.images) if f.lower().endswith((".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff")))
        if len(names) < n_total:
            raise SystemExit(f"--images {args.images}: {len(names)} images, {n_total} needed (--batch x --gpus)")
        user_images = [os.path.join(args.images, f) for f in names[:n_total]]
        nt = preprocess_lines([read_rgb(f) for f in user_images[lo:hi]], device=dev)
        x, mask = nt.tensors, nt.mask
        canvas_w = int(x.shape[3])

        def to_batch(lines):                                  # noqa: F811
            t = preprocess_lines(lines, device=dev)
            if t.tensors.shape[2:] != x.shape[2:]:            # the dp self-check compares records only: any canvas is a valid padding of the same lines
                log(f"shard canvas {tuple(t.tensors.shape[2:])} != rank 0's {tuple(x.shape[2:])}")
            return t.tensors, t.mask

        def make_lines(a, b):                                 # noqa: F811
            return [read_rgb(f) for f in user_images[a:b]]
    elif evalshape:
        # the reference's eval pipeline on the device: uint8 lines -> resize (Pillow's fixed-point bilinear, bit-exact) -> /255 -> normalise
        # -> pad + mask, ONE launch (dtlr_preprocess_lines); the timed step then runs on the resulting 83x1328 canvas
        from dtlr_amd.transforms import preprocess_lines
        raw = synth.uint8_lines(hi - lo, args.height, widths[lo:hi], seed=1000, start=lo)
        nt = preprocess_lines(raw, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            nt = preprocess_lines(raw, device=dev)
        torch.cuda.synchronize()
        preproc_ms = (time.perf_counter() - t0) / 5 * 1e3
        x, mask = nt.tensors, nt.mask

        def to_batch(lines):                                  # noqa: F811  (dp self-check of this config: other shards through the same transform)
            t = preprocess_lines(lines, device=dev)
            return t.tensors, t.mask

        def make_lines(a, b):                                 # noqa: F811
            return synth.uint8_lines(b - a, args.height, widths[a:b], seed=1000, start=a)
    else:
        imgs = make_lines(lo, hi)
        x, mask = to_batch(imgs)
    padded = chinese or mixed or bool(mask.any().item())
    observed = None
    if args.weights or args.images:
        # what the synthetic generator only assumes: the backbone's activation peak (fp16 / split engines need < 65504)
        # and how far the encoder's offset heads look (the LDS-window sampler's far fraction at a halo of 8 columns, per layer after calibration)
        f_, l_, _hw = eng.features(x)
        observed = {"backbone_activation_peak": float(max(t.float().abs().max() for t in list(f_) + [l_])), "canvas": [int(x.shape[2]), int(x.shape[3])]}
        del f_, l_

    # --single-device: all ranks drive cuda:0 (the 2-ranks-on-one-GPU test; gloo, because RCCL refuses duplicate devices).  Their kernels
    # interleave freely on the device -- round 2 serialised the ranks through a file lock here because ~10% of forwards then differed;
    # the cause (a lane-mask sequence of the decoder's deformable-sampling kernel) is fixed and the lock is gone.
    def local_step(xx=None, mm=None, debug=False):
        out = eng.forward(x if xx is None else xx, mask if mm is None else mm, has_padding=padded, return_debug=debug)
        rec = decode_blank_records(out)
        return (rec, out) if debug else rec

    def step():
        labels, lengths = local_step()
        return ddist.all_gather_records(labels, lengths, n_total)

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")

    def timed_block():
        ddist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r = step()
        torch.cuda.synchronize()
        ddist.barrier()
        return ddist.max_over_ranks(time.perf_counter() - t0, dev), r

    ops.MSDA_EVENTS = []
    first, rec = timed_block()
    blocks = [first]
    # every rank derives the same repeat count from the max-reduced first block
    repeats = 1 if first >= args.min_seconds else min(50, int(args.min_seconds / max(first, 1e-6)) + 1)
    for _ in range(repeats - 1):
        t, rec = timed_block()
        blocks.append(t)
    events, ops.MSDA_EVENTS = ops.MSDA_EVENTS, None
    elapsed = sorted(blocks)[len(blocks) // 2]
    log(f"timed {len(blocks)} blocks of {args.steps} steps: median {elapsed:.4f}s (min {min(blocks):.4f}, max {max(blocks):.4f})")

    # ---- data-parallel self-check: rank 0 recomputes every shard and compares with the gathered records, bit for bit ----
    dp_verified = None
    if world > 1 and rank == 0:
        ok = True
        for r in range(world):
            a, b = ddist.shard_bounds(n_total, r, world)
            xs, ms = (x, mask) if r == 0 else to_batch(make_lines(a, b))
            lab, ln = local_step(xs, ms)
            same = bool(torch.equal(lab.cpu(), rec[0][a:b].cpu()) and torch.equal(ln.cpu(), rec[1][a:b].cpu()))
            if not same:
                diff = (lab.cpu() != rec[0][a:b].cpu())
                log(f"dp self-check: shard {r} differs: {int(diff.sum())} label slots in lines {sorted(set(diff.nonzero()[:, 0].tolist()))}, "
                    f"lengths equal: {bool(torch.equal(ln.cpu(), rec[1][a:b].cpu()))}")
            ok &= same
        dp_verified = ok
        log(f"dp_verified = {ok}")

    # MFMA-class launches are timed in a REPLAY of the same steps right after the timed region (same inputs, same stream)
    mfma_events = []
    replay_steps = min(args.steps, 5)
    if rank == 0:
        ops.MFMA_EVENTS = mfma_events
        for _ in range(replay_steps):
            local_step()                                        # no collective here: the other ranks have left the timed region
        torch.cuda.synchronize()
        ops.MFMA_EVENTS = None

    if rank != 0:
        ddist.finalize()                 # waits for rank 0 (replay + printing) at a last barrier, then tears the group down
        return
    total_steps = args.steps * len(blocks)
    enc = [(a.elapsed_time(b), n, lq, s) for (a, b, n, lq, s) in events if lq == s]
    dec = [(a.elapsed_time(b), n, lq, s) for (a, b, n, lq, s) in events if lq != s]
    velem = 2 if dtype in (torch.bfloat16, torch.float16) else 4
    traffic_db, traffic_file = {}, None
    for fn in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        if fn.endswith("_traffic.json"):
            try:
                traffic_db.update(json.load(open(os.path.join(ROOT, "profiles", fn))))
                traffic_file = fn
            except Exception:
                pass
    # PMC traffic figures were measured on the Latin bf16 B = 32 step (profiles/*_traffic.json): attach them to that configuration only
    traffic_ok = args.config == "latin" and not args.backbone and B == 32 and args.dtype == "bf16" and canvas_w == 2048 and args.height == 128
    roof = None
    if enc:
        ms = sum(e[0] for e in enc) / len(enc)
        n, lq, s = enc[0][1], enc[0][2], enc[0][3]
        alg = msda_algorithmic_bytes_per_line(s, lq, velem) * n
        achieved = alg / (ms * 1e-3) / 1e9
        traffic = traffic_db.get(f"{args.dtype}_enc_bytes_per_launch") if traffic_ok else None
        roof = {"bound": "hbm", "kernel": f"msda_enc_lds_kernel (deformable sampling, encoder call, Lq=S={s}/line)",
                "symbol": "msda_enc_lds_kernel<unsigned short, unsigned short, 2, 512>" if velem == 2 else "msda_enc_lds_kernel<float, float, 3, 512>",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg, "mean_launch_ms": round(ms, 4), "launches_timed": len(enc)}
        if dec:
            msd = sum(e[0] for e in dec) / len(dec)
            n, lq, s = dec[0][1], dec[0][2], dec[0][3]
            algd = msda_algorithmic_bytes_per_line(s, lq, velem, ref_dim=4) * n
            roof["decoder_call"] = {"achieved": round(algd / (msd * 1e-3) / 1e9, 1), "mean_launch_ms": round(msd, 4)}
    # ---- per-class rooflines; `roofline` = the class with the largest share of the timed region ----
    by_kernel = []
    if roof:
        r = dict(roof)
        r["ms_per_step"] = round(sum(e[0] for e in enc) / total_steps, 3)
        by_kernel.append(r)
    classes, shapes = {}, {}
    for ev in mfma_events:
        a, b, kind, flops, nbytes = ev[:5]
        tag = ev[5] if len(ev) > 5 else None
        sym = ev[6] if len(ev) > 6 else None
        dt = a.elapsed_time(b)
        # a launch that is ONE kernel is accounted under that kernel's symbol (and its token count: the encoder and decoder calls of a
        # kernel are different launches in the rocprof summary's average, so they stay together here too); the rest per class
        c = classes.setdefault((kind, sym), [0.0, 0.0, 0, 0.0])
        c[0] += dt; c[1] += flops; c[2] += 1; c[3] += nbytes
        if tag and kind.startswith("gemm"):
            s_ = shapes.setdefault((kind, tag), [0.0, 0.0, 0, 0.0])
            s_[0] += dt; s_[1] += flops; s_[2] += 1; s_[3] += nbytes
    names = {"gemm_bf16": "gemm_ws_kernel<bf16> / gemm_k256_kernel / gemm_kres_kernel / conv3x3_patch_kernel (every Linear / 1x1 conv / 3x3 conv, fused epilogues)",
             "gemm_f32": "gemm_ws_kernel<f32> (fp32 heads and selection scores, exact-fp32 MFMA 16x16x4)",
             "gemm_f32s": "gemm_ws_kernel<f32s> (fp32 operands as fp16 hi + lo halves, three 16x16x32 fp16 MFMAs per product; flops counted once)",
             "ffn_fused_bf16": "ffn3_bf16_kernel / ffn2_bf16_kernel / ffn_fused_bf16_kernel (linear1+ReLU+linear2+residual+LayerNorm, intermediate on chip)",
             "ffn_fused_f32s": "ffn_split_kernel (linear1+ReLU+linear2+residual+LayerNorm on split fp16 hi + lo operands, intermediate on chip; flops counted once)",
             "proj_ln_bf16": "proj_ln_bf16_kernel (attention output projection + residual + LayerNorm)"}

    def both_roofs(kind, ms, flops, nbytes):
        # f32s: algorithmic flops (each product once) against a third of the dense 16-bit peak -- the kernel issues three MFMAs per product
        peak = MFMA_PEAK_F32_TFLOPS if kind == "gemm_f32" else (MFMA_PEAK_BF16_TFLOPS / 3.0 if kind.endswith("f32s") else MFMA_PEAK_BF16_TFLOPS)
        ach = flops / (ms * 1e-3) / 1e12
        gbps = nbytes / (ms * 1e-3) / 1e9                        # compulsory operand + result bytes (each tensor once)
        return peak, ach, gbps, ach / peak, gbps / HBM_PEAK_GBS

    for (kind, sym), (ms, flops, cnt, nbytes) in classes.items():
        peak, ach, gbps, mf, hf = both_roofs(kind, ms, flops, nbytes)
        # SURVEY 8(d) labels every GEMM "MFMA-bound"; at K = 256 (128 flop/byte < 312) the binding roof is HBM.  Both fractions are
        # reported; `bound`/`achieved`/`peak`/`frac` name the larger (binding) one, `mfma_frac` is the 8(d) label's number.
        label = (sym + ": " if sym else "") + names.get(kind, kind)
        if hf > mf:
            head = {"bound": "hbm", "kernel": label, "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hf, 4)}
        else:
            head = {"bound": "mfma", "kernel": label, "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(mf, 4)}
        head["symbol" if sym else "class"] = sym if sym else True
        by_kernel.append({**head, "traffic": ((traffic_db.get(f"symbol:{sym}") or {}).get("bytes_per_dispatch_mean") if sym else traffic_db.get(f"{kind}_bytes_per_launch_mean")) if traffic_ok else None,
                          "mfma_tflops": round(ach, 1), "mfma_frac": round(mf, 4),
                          "hbm_gbps_algorithmic": round(gbps, 1), "hbm_frac": round(hf, 4),
                          "algorithmic_flops_per_launch": round(flops / cnt), "algorithmic_bytes_per_launch": round(nbytes / cnt),
                          "mean_launch_ms": round(ms / cnt, 4), "launches_timed": cnt, "ms_per_step": round(ms / replay_steps, 3),
                          "timed_in": "replay of the timed steps, launches >= 2 GFLOP only"})
    by_kernel.sort(key=lambda r: -r["ms_per_step"])
    gemm_by_shape = []
    for (kind, tag), (ms, flops, cnt, nbytes) in shapes.items():
        peak, ach, gbps, mf, hf = both_roofs(kind, ms, flops, nbytes)
        gemm_by_shape.append({"shape": tag, "kind": kind, "launches_per_step": round(cnt / replay_steps, 1), "mean_launch_us": round(1e3 * ms / cnt, 1),
                              "ms_per_step": round(ms / replay_steps, 3), "mfma_tflops": round(ach, 1), "mfma_frac": round(mf, 4),
                              "hbm_gbps_algorithmic": round(gbps, 1), "hbm_frac": round(hf, 4), "traffic": traffic_db.get(f"gemm:{tag}") if traffic_ok else None})
    gemm_by_shape.sort(key=lambda r: -r["ms_per_step"])
    # `roofline` = the single KERNEL with the largest share of the step (class rows aggregate many instantiations: never the headline)
    dominant = next((r for r in by_kernel if r.get("symbol")), by_kernel[0] if by_kernel else None)
    line = {
        "metric": ("text-lines/sec (Chinese 7356-class head, mixed-length 128x2560, bs=32)" if chinese else
                   "text-lines/sec (Latin, mixed widths padded to 128x2048, bs=32)" if mixed else
                   f"text-lines/sec (Latin, eval transform: 128x2048 uint8 lines -> {x.shape[2]}x{x.shape[3]} canvases, bs=32)" if evalshape else
                   "text-lines/sec (128x2048, bs=32)"),
        "value": round(n_total * args.steps / elapsed, 2), "unit": "lines/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic" if not args.images else f"user images ({os.path.basename(os.path.normpath(args.images))})",
        "timed_blocks": len(blocks), "block_ms": [round(b * 1e3, 2) for b in blocks],
        "config": {"workload": (f"Latin DTLR (C=166) forward+decode, {B} synthetic lines per GPU, widths seeded from {{{canvas_w - 768}..{canvas_w}}}, zero-padded to {args.height}x{canvas_w} with masks"
                                if mixed else
                                f"Latin DTLR (C=166) forward+decode on the eval-transformed canvas ({x.shape[2]}x{x.shape[3]}; preprocessing {preproc_ms:.3f} ms per batch, outside the timed step), {B} synthetic uint8 {args.height}x{canvas_w} lines per GPU"
                                if evalshape else
                                f"Latin DTLR (ResNet-50 + 6/6 deformable DETR, C=166) forward+decode, {B} synthetic {args.height}x{canvas_w} lines per GPU"
                                if not chinese else
                                f"Chinese DTLR (C=7356) forward+decode, {B} synthetic lines per GPU, widths seeded from {{{canvas_w - 1024}..{canvas_w}}}, zero-padded to {args.height}x{canvas_w} with masks")
                               + f", random-init name-seeded weights (generator v{weights.GENERATOR_VERSION})",
                   "weights": args.weights or f"synthetic, generator v{weights.GENERATOR_VERSION}", "images": args.images or "synthetic noise lines (seed 1000)",
                   "canvas": [int(x.shape[2]), int(x.shape[3])], "num_classes": cfg.num_classes,
                   "backbone": cfg.backbone, "engine_opts": args.engine_opt, "global_batch": n_total, "parallelism": f"dp{world}",
                   "library_backed_ops": sorted(ops.LIBRARY_BACKED)},
        "distributed": {"backend": tdist.get_backend() if (tdist.is_available() and tdist.is_initialized()) else None,
                        "world_size": tdist.get_world_size() if (tdist.is_available() and tdist.is_initialized()) else 1,
                        # ranks in the RCCL communicator as torch.distributed reports them (backend "nccl" IS RCCL on ROCm): a first multi-GPU run describes itself
                        "rccl_ranks": (tdist.get_world_size() if (tdist.is_available() and tdist.is_initialized() and tdist.get_backend() == "nccl") else 0),
                        "devices_visible": torch.cuda.device_count(),
                        "dp_verified": dp_verified, "host_placement": placement},
        "roofline": dominant,
        # `traffic` figures are PMC measurements looked up from profiles/ (not taken in this run): the file and the commit they were measured at
        "traffic_source": {"file": traffic_file, "measured_at_commit": traffic_db.get("_commit"), "attached": bool(traffic_ok)},
        "roofline_by_kernel": by_kernel,
        "gemm_by_shape": gemm_by_shape[:24],
    }
    if world == 1 and args.dtype in ("bf16", "f16") and not cfg.is_swin:
        try:                                                    # the MSDA figure above is the synthetic weights' best case: show the other cases beside it
            lhw = [tuple(int(v) for v in hw) for hw in eng._shape_cache[next(iter(eng._shape_cache))]["level_hw"]] if eng._shape_cache else None
            if lhw is None:
                lhw = [(16, canvas_w // 8), (8, canvas_w // 16), (4, canvas_w // 32), (2, canvas_w // 64)]
            line["msda_encoder_by_offset_scale"] = msda_offset_sensitivity(eng, B, dtype, lhw)
            log(f"msda_encoder_by_offset_scale: {line['msda_encoder_by_offset_scale']}")
        except Exception as e:
            line["msda_encoder_by_offset_scale"] = {"error": repr(e)}
    if world == 1 and not args.no_bs1:
        # the reference's evaluation loop feeds ONE line at a time (/root/reference/evaluation.py:494-499): forward + decode latency of a single
        # line of the batch, and of 4 (the decoder's ~60 launches do not shrink with the batch)
        lat = {}
        for nb in (1, 4):
            if nb > B:
                continue
            xs, ms_ = x[:nb].contiguous(), mask[:nb].contiguous()
            for _ in range(3):
                local_step(xs, ms_)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                local_step(xs, ms_)
            torch.cuda.synchronize()
            lat[nb] = (time.perf_counter() - t0) / 20 * 1e3
        line["latency_ms_bs1"] = round(lat[1], 3)
        line["latency_ms_by_batch"] = {str(k): round(v, 3) for k, v in lat.items()}
        log(f"single-line latency: {lat}")
        # the same single line as a HIP-graph replay (round 6: one line is ~200 dependent launches; the replay is bit-identical to the eager
        # forward -- tests/test_gpu_model.py -- and, unlike at 4 or 32 lines, measurably shorter)
        try:
            xs, ms_ = x[:1].contiguous(), mask[:1].contiguous()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    local_step(xs, ms_)
            torch.cuda.current_stream().wait_stream(side)
            eager_rec = [t.clone() for t in local_step(xs, ms_)]
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                graph_rec = local_step(xs, ms_)
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                graph.replay()
            torch.cuda.synchronize()
            line["latency_ms_bs1_graph"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
            line["latency_bs1_graph_equals_eager"] = bool(all(torch.equal(a, b) for a, b in zip(graph_rec, eager_rec)))
            log(f"single-line latency, HIP-graph replay: {line['latency_ms_bs1_graph']} ms (records == eager: {line['latency_bs1_graph_equals_eager']})")
            del graph
        except Exception as e:
            line["latency_ms_bs1_graph"] = None
            log(f"single-line HIP-graph leg failed: {e!r}")
    rows = sorted({int(round(i * (B - 1) / max(args.parity_lines - 1, 1))) for i in range(min(args.parity_lines, B))})
    ob = None
    if observed is not None:
        observed["msda_encoder_choice_by_layer"] = {k[0]: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                                                    for k, v in eng._msda_state.items()}      # mode / halo / far fraction the calibration saw
        line["observed_on_user_assets"] = observed
    if not args.no_parity:
        try:
            from oracle.parity import OracleBatch          # the checker: never the product path
            t0 = time.perf_counter()
            ob = OracleBatch(cfg, sd, x[rows], mask[rows])
            log(f"oracle free run of {len(rows)} lines: {time.perf_counter() - t0:.1f}s")
            (_, out) = local_step(debug=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            line["parity_vs_oracle"] = parity_vs_oracle(ob, args.dtype, out, rows, chinese)
            line["parity_vs_oracle"]["rows"] = rows
            line["parity_vs_oracle"]["weights"] = f"generator v{weights.GENERATOR_VERSION} (the benched weights; characters planted per selection rank)" if not args.weights else args.weights
            log(f"parity_vs_oracle ({time.perf_counter() - t0:.1f}s): {line['parity_vs_oracle']}")
            del out
        except Exception as e:                                   # the parity leg must never cost the throughput line
            line["parity_vs_oracle"] = {"error": repr(e)}
    # ---- free-running leg on generator-v4 weights (Latin only: the generator's calibration), every engine against ONE oracle run
    ob4 = sd4 = None
    rows4 = sorted({int(round(i * (B - 1) / max(V4_PARITY_LINES - 1, 1))) for i in range(min(V4_PARITY_LINES, B))})
    if not args.no_parity and args.config == "latin" and not args.backbone and not args.weights and not args.images:
        try:
            from oracle.parity import OracleBatch
            t0 = time.perf_counter()
            sd4 = weights.synthetic_state_dict(cfg, seed=0, version=4)
            ob4 = OracleBatch(cfg, sd4, x[rows4], mask[rows4])
            log(f"oracle free run, generator v4, {len(rows4)} lines: {time.perf_counter() - t0:.1f}s; characters per line {[len(s_) for s_ in ob4.strings]}")
            line["free_running_v4"] = free_running_v4(cfg, sd4, ob4, args.dtype, dtype, dev, x, mask, rows4, padded)
            log(f"free_running_v4: {line['free_running_v4']}")
        except Exception as e:
            line["free_running_v4"] = {"error": repr(e)}
    # ---- the other two engines on the SAME batch: a short timed block each + the same parity leg.  `value` above is --dtype's. ----
    by_dtype = {args.dtype: {"lines_per_s": line["value"], "ms_per_step": line["ms_per_step"], "steps_timed": total_steps,
                             "parity_vs_oracle": line.get("parity_vs_oracle"), "free_running_v4": line.get("free_running_v4")}}
    if world == 1 and not args.no_other_dtypes:
        del eng
        torch.cuda.empty_cache()
        for name in ("bf16", "f16", "f32s", "f32"):
            if name == args.dtype:
                continue
            try:
                e2 = DTLREngine(cfg, sd, dev, DT[name], split=name == "f32s")

                def step2(debug=False):
                    o = e2.forward(x, mask, has_padding=padded, return_debug=debug)
                    return (decode_blank_records(o), o)
                for _ in range(2):
                    step2()
                torch.cuda.synchronize()
                k2 = args.steps if name in ("bf16", "f16") else max(3, args.steps // (2 if name == "f32s" else 5))
                t0 = time.perf_counter()
                for _ in range(k2):
                    step2()
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t0
                ent = {"lines_per_s": round(B * k2 / dt2, 2), "ms_per_step": round(dt2 / k2 * 1e3, 3), "steps_timed": k2}
                if ob is not None:
                    _, o = step2(debug=True)
                    torch.cuda.synchronize()
                    ent["parity_vs_oracle"] = parity_vs_oracle(ob, name, o, rows, chinese)
                    del o
                if ob4 is not None:
                    del e2
                    e2 = None
                    torch.cuda.empty_cache()
                    ent["free_running_v4"] = free_running_v4(cfg, sd4, ob4, name, DT[name], dev, x, mask, rows4, padded)
                by_dtype[name] = ent
                log(f"engine {name}: {ent}")
                del e2
                torch.cuda.empty_cache()
            except Exception as e:
                by_dtype[name] = {"error": repr(e)}
    line["by_dtype"] = by_dtype
    if world == 1 and not args.no_cpu_baseline and args.config == "latin" and not args.backbone:
        line["cpu_baseline"] = cpu_baseline()
        line["speedup_vs_cpu"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
    line["detail_file"] = write_detail(line, args.detail)
    log("full result (also in " + (line["detail_file"] or "<not written>") + "): " + json.dumps(line))
    print(json.dumps(compact_line(line)), flush=True)          # ONE line, < 6 KB (STDOUT_LINE_LIMIT)
    ddist.finalize()


if __name__ == "__main__":
    main()
