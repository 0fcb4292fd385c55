#!/usr/bin/env python3
"""Static resource table of every kernel in dtlr_amd/csrc (no GPU): VGPRs, AGPRs, SGPR / VGPR spills, scratch bytes, LDS, occupancy, as
hipcc's -Rpass-analysis=kernel-resource-usage reports them for gfx950.
    python tools/resource_usage.py [--defs=-DDTLR_HALF_IS_F16] [--min-vgprs 0] > profiles/rNN_kernel_resources.txt
A kernel with scratch > 0 spills registers to memory: none of the hot kernels may (the last column flags it)."""
import argparse
import concurrent.futures as cf
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["VGPRs", "AGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"]


def one(src, defs):
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/dtlr_amd/csrc",
               "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(d, "x.o")] + defs
        out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for l in out.split("\n"):
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = {"name": m.group(1), "file": os.path.basename(src)}
            rows.append(cur)
            continue
        for k in KEYS:
            m = re.search(r"remark:\s+" + re.escape(k) + r": (\d+)", l)
            if m and cur is not None:
                cur[k] = int(m.group(1))
    return rows


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [re.sub(r"\(.*", "", o).replace("void dtlr::", "").replace("dtlr::", "") for o in out[: len(names)]]
    except Exception:
        return names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--defs", default="")
    ap.add_argument("--min-vgprs", type=int, default=0)
    args = ap.parse_args()
    srcs = sorted(glob.glob(os.path.join(ROOT, "dtlr_amd", "csrc", "*.hip")))
    with cf.ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        rows = [r for rs in ex.map(lambda s: one(s, args.defs.split()), srcs) for r in rs]
    names = demangle([r["name"] for r in rows])
    print(f"# gfx950 kernel resources ({len(rows)} kernels; hipcc -O3 {args.defs}): VGPR AGPR spill(S/V) scratch LDS(static) occupancy")
    for r, n in sorted(zip(rows, names), key=lambda x: (x[0]["file"], x[1])):
        if r.get("VGPRs", 0) < args.min_vgprs:
            continue
        flag = "  <-- SPILLS" if r.get("ScratchSize [bytes/lane]", 0) or r.get("VGPRs Spill", 0) else ""
        print(f"{r['file']:16s} {n[:86]:86s} v{r.get('VGPRs', 0):4d} a{r.get('AGPRs', 0):4d}  sp {r.get('SGPRs Spill', 0)}/{r.get('VGPRs Spill', 0)}"
              f"  scratch {r.get('ScratchSize [bytes/lane]', 0):4d}  lds {r.get('LDS Size [bytes/block]', 0):6d}  occ {r.get('Occupancy [waves/SIMD]', 0)}{flag}")


if __name__ == "__main__":
    main()
