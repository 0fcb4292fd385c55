#!/usr/bin/env python3
"""Hazard lint for the inline-asm MFMAs of dtlr_amd/csrc/ffn4.hip (no GPU needed).

The kernel issues its MFMAs as inline asm (tied accumulator operands: see the file header), so hipcc's hazard recogniser inserts no wait
states for them.  The one hazard the surrounding compiler-generated code can create is a VALU write of a register that an MFMA reads as
SrcA / SrcB / SrcC a few instructions later (e.g. a v_mov that re-assembles an operand tuple).  This script compiles the file, walks the
kernel's assembly and reports every v_mfma whose source registers were written by a VALU instruction fewer than WAIT wait states earlier
(s_nop N counts N + 1, any other instruction 1).   python tools/ffn4_lint.py [--defs=-DDTLR_HALF_IS_F16]"""
import argparse, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WAIT = 3


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def lint(defs=""):
    src = os.path.join(ROOT, "dtlr_amd", "csrc", "ffn4.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "ffn4.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "--cuda-device-only", "-S", "-o", out, src] + defs.split()
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        lines = [l.strip() for l in open(out)]
    ins = [l for l in lines if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
    bad, nm = [], 0
    for i, l in enumerate(ins):
        if not l.startswith("v_mfma"):
            continue
        nm += 1
        ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
        srcs = set().union(*[regs(t) for t in ops[1:]])
        ws, k = 0, i - 1
        while k >= 0 and ws < WAIT:
            p = ins[k]
            op = p.split()[0]
            if op == "s_nop":
                ws += int(p.split()[1]) + 1
            else:
                if op.startswith("v_") and not op.startswith(("v_mfma", "v_cmp", "v_accvgpr_write", "v_accvgpr_mov")):
                    dst = regs(p.split(None, 1)[1].split(",")[0].strip()) if len(p.split(None, 1)) > 1 else set()
                    if dst & srcs:
                        bad.append((l, p, ws))
                ws += 1
            k -= 1
    return nm, bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--defs", default="")
    a = ap.parse_args()
    n, bad = lint(a.defs)
    print(f"{n} MFMAs checked, {len(bad)} with a VALU-written source fewer than {WAIT} wait states earlier")
    for l, p, ws in bad[:20]:
        print("   ", p, " ->", l, f"({ws} wait states)")
    sys.exit(1 if bad else 0)
