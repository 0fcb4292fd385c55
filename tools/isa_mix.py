#!/usr/bin/env python3
"""Static instruction mix of a kernel's loops from the gfx950 assembly hipcc emits (no GPU needed).
    python tools/isa_mix.py dtlr_amd/csrc/msda_enc.hip msda_enc_lds_kernelIttLi1ELi512E [--defs=-DDTLR_HALF_IS_F16] [--top 40]
Compiles the file with --save-temps into a scratch directory, finds the first kernel whose mangled name contains the pattern,
lists its loops (backward branches) with VALU / SALU / LDS / VMEM counts and prints the opcode histogram of the longest one.
For a VALU-issue-bound kernel (msda_enc_lds_kernel: SQ counters put VALU issue at ~96% of its duration) the VALU count of the
main loop IS the cost model: an edit that removes 10% of it is worth ~10% of the kernel."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def category(op: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("pattern", help="substring of the mangled kernel name")
    ap.add_argument("--defs", default="", help="extra compiler flags, e.g. -DDTLR_HALF_IS_F16")
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    src = os.path.abspath(args.source)
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/dtlr_amd/csrc",
               "--save-temps", "-c", src, "-o", "x.o"] + args.defs.split()
        subprocess.check_call(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(d) if f.endswith(".s") and "amdgcn" in f]
        lines = open(os.path.join(d, asm[0])).read().split("\n")
    starts = [i for i, l in enumerate(lines) if args.pattern in l and l.rstrip().split(";")[0].rstrip().endswith(":") and not l.startswith((".", "\t"))]
    if not starts:
        sys.exit(f"no kernel matching {args.pattern!r}")
    start = starts[0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    print(f"kernel {lines[start].split(':')[0]}: {len(body)} assembly lines")
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i, m.group(1)))

    def mix(a, b):
        cats, ops = {}, {}
        for l in body[a:b + 1]:
            l = l.split(";")[0].strip()
            if not l or l.startswith(".") or l.endswith(":"):
                continue
            op = l.split()[0]
            cats[category(op)] = cats.get(category(op), 0) + 1
            ops[op] = ops.get(op, 0) + 1
        return cats, ops

    loops.sort(key=lambda x: x[0] - x[1])
    for a, b, t in loops[:8]:
        print(f"  loop {t}: lines {a}..{b}  {mix(a, b)[0]}")
    if not loops:
        loops = [(0, len(body) - 1, "the whole kernel (no loop)")]
        print(f"  {mix(0, len(body) - 1)[0]}")
    if loops:
        a, b, t = loops[0]
        print(f"opcode histogram of {t} (both sides of every branch inside it are counted):")
        for op, n in sorted(mix(a, b)[1].items(), key=lambda x: -x[1])[: args.top]:
            print(f"  {op:30s}{n}")


if __name__ == "__main__":
    main()
