#!/usr/bin/env python3
"""Which kernels of a .hip file changed between a git revision and the working tree -- at the instruction level (no GPU).
    python tools/isa_diff.py dtlr_amd/csrc/msda_enc.hip [--rev HEAD] [--defs=-DDTLR_HALF_IS_F16]
Compiles both versions for gfx950 with --save-temps and compares every kernel's instruction stream (basic-block label numbers
normalised).  Use: adding a variant / template instantiation next to a verified kernel must leave the verified kernel SAME."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(src_text, name, defs):
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, name), "w").write(src_text)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/dtlr_amd/csrc",
                               "--save-temps", "-c", name, "-o", "x.o"] + defs, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(d) if f.endswith(".s") and "amdgcn" in f][0]
        lines = open(os.path.join(d, asm)).read().split("\n")
    out, i = {}, 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", lines[i])
        if m and i + 1 < len(lines) and not lines[i].startswith("."):
            j = i
            while j < len(lines) and not lines[j].strip().startswith("s_endpgm"):
                if j > i and re.match(r"^(_Z\w+):", lines[j]):
                    break
                j += 1
            if j < len(lines) and lines[j].strip().startswith("s_endpgm"):
                out[m.group(1)] = [re.sub(r"\.LBB\d+_", ".LBB_", x.split(";")[0].rstrip()) for x in lines[i + 1:j + 1] if x.split(";")[0].strip()]
                i = j
        i += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--rev", default="HEAD")
    ap.add_argument("--defs", default="")
    args = ap.parse_args()
    rel = os.path.relpath(os.path.abspath(args.source), ROOT)
    old = subprocess.check_output(["git", "show", f"{args.rev}:{rel}"], cwd=ROOT, text=True)
    new = open(os.path.join(ROOT, rel)).read()
    a, b = kernels(old, os.path.basename(rel), args.defs.split()), kernels(new, os.path.basename(rel), args.defs.split())
    changed = 0
    for k in sorted(set(a) | set(b)):
        if k not in a:
            print(f"NEW      {k}  ({len(b[k])} instructions)")
        elif k not in b:
            print(f"REMOVED  {k}")
            changed += 1
        elif a[k] == b[k]:
            print(f"SAME     {k}  ({len(a[k])})")
        else:
            print(f"CHANGED  {k}  ({len(a[k])} -> {len(b[k])})")
            changed += 1
    sys.exit(1 if changed else 0)


if __name__ == "__main__":
    main()
