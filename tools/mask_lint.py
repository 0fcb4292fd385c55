#!/usr/bin/env python3
"""Static lint for the round-2/3 contention bug: the decoder's deformable-sampling kernel kept 64-bit lane masks --
results of s_and_b64 / s_or_b64 / v_cmp ... s[a:b] chains -- live in SGPR pairs across its gather (an `s_waitcnt vmcnt` many hundred cycles
later), and under multi-process contention the upper 16 lanes of such a mask were occasionally lost (whole heads of odd queries scaled by
0.87..0.98: dropped corner terms).  The fix removed the long-lived masks from that kernel; this tool reports, for EVERY kernel of the
library, how many SALU / VALU-compare produced lane masks in SGPR pairs are still read after an intervening `s_waitcnt vmcnt(..)`, so that
a kernel with the same exposure is visible before it fails.  Pure text analysis of hipcc's gfx950 assembly, no GPU:

    python tools/mask_lint.py [--defs=-DDTLR_HALF_IS_F16] > profiles/r04_mask_lint.txt

Approximations (a lint, not a proof): the assembly is scanned linearly (control flow ignored: a mask defined before a loop and used inside
it is counted once), `exec` itself is excluded (saved / restored exec masks of divergent branches are listed separately: `exec_saves`),
and a use is any instruction that names the pair as a source (v_cndmask, s_and / s_or / s_andn2, s_and_saveexec, v_cmp writing back)."""
import argparse
import concurrent.futures as cf
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASK_DEF = re.compile(r"^(s_(?:and|or|xor|andn2|orn2|nand|nor|xnor|cselect|mov)_b64|v_cmp\w*_e64|v_cmpx?\w*)\s+(s\[\d+:\d+\]|vcc)\b")
PAIR = re.compile(r"s\[(\d+):(\d+)\]")
WAIT_VM = re.compile(r"^s_waitcnt\b.*vmcnt\(")


def kernels_of(src, defs):
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/dtlr_amd/csrc",
               "--save-temps", "-c", src, "-o", "x.o"] + defs
        subprocess.check_call(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(d) if f.endswith(".s") and "amdgcn" in f]
        lines = open(os.path.join(d, asm[0])).read().split("\n")
    out, cur, name = [], None, None
    for l in lines:
        if l and not l.startswith((".", "\t", " ", ";")) and l.rstrip().split(";")[0].rstrip().endswith(":") and l.startswith("_Z"):
            name, cur = l.split(":")[0], []
            continue
        if cur is not None:
            t = l.split(";")[0].strip()
            if t.startswith("s_endpgm"):
                out.append((name, cur))
                cur = None
            elif t and not t.startswith(".") and not t.endswith(":"):
                cur.append(t)
    return out


def lint(body):
    """-> (masks defined, SGPR-pair masks read after a vmcnt wait, longest such def->use distance in instructions, exec saves live across a
    vmcnt wait, VCC masks read after a vmcnt wait)"""
    live = {}           # pair -> (def index, waits seen since def)
    n_def = n_bad = far = exec_bad = vcc_bad = salu_ops = cnd_sgpr = 0
    counted = set()
    for i, t in enumerate(body):
        op = t.split()[0]
        if WAIT_VM.match(t):
            for k in live:
                live[k][1] += 1
            continue
        args = t[len(op):]
        if re.match(r"s_(and|or|xor|andn2|orn2|nand|nor|xnor)_b64$", op) and "exec" not in args:
            salu_ops += 1                   # mask ARITHMETIC (combining compare results): the failing kernel had 37 of these per thread
        if op.startswith("v_cndmask") and PAIR.search(args.rsplit(",", 1)[-1]):
            cnd_sgpr += 1                   # a select driven by a mask held in an SGPR pair (25 in the failing kernel)
        m = MASK_DEF.match(t)
        dst = None
        if m:
            dst = m.group(2)
            srcs = args.split(",", 1)[1] if "," in args else ""
        else:
            srcs = args
            if op.startswith("s_and_saveexec") or op.startswith("s_or_saveexec"):
                d = PAIR.search(args)
                if d:
                    live["x" + d.group(0)] = [i, 0]
        for p in set(PAIR.findall(srcs)) | ({("vcc", "")} if re.search(r"\bvcc\b", srcs) else set()):
            key = "vcc" if p[0] == "vcc" else f"s[{p[0]}:{p[1]}]"
            for kk, is_exec in ((key, False), ("x" + key, True)):
                if kk in live and live[kk][1] > 0 and (kk, live[kk][0]) not in counted:
                    counted.add((kk, live[kk][0]))
                    if is_exec:
                        exec_bad += 1
                    elif kk == "vcc":
                        vcc_bad += 1        # v_cmp -> VCC -> v_cndmask with a wait scheduled in between: one pair, rewritten every few instructions
                    else:
                        n_bad += 1
                        far = max(far, i - live[kk][0])
        if dst is not None:
            n_def += 1
            live[dst] = [i, 0]
    return n_def, n_bad, far, exec_bad, vcc_bad, salu_ops, cnd_sgpr


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o).replace("void dtlr::", "").replace("dtlr::", "") for o in out[: len(names)]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--defs", default="")
    ap.add_argument("--files", default="", help="comma-separated basenames (default: every csrc/*.hip)")
    args = ap.parse_args()
    srcs = sorted(glob.glob(os.path.join(ROOT, "dtlr_amd", "csrc", "*.hip")))
    if args.files:
        srcs = [s for s in srcs if os.path.basename(s) in args.files.split(",")]
    with cf.ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        per_file = list(ex.map(lambda s: kernels_of(s, args.defs.split()), srcs))
    rows = []
    for src, ks in zip(srcs, per_file):
        names = demangle([k[0] for k in ks])
        for (mangled, body), nm in zip(ks, names):
            rows.append((os.path.basename(src), nm, len(body)) + lint(body))
    print(f"# lane masks in SGPR pairs read after an `s_waitcnt vmcnt` ({len(rows)} kernels, hipcc -O3 gfx950 {args.defs}); see tools/mask_lint.py")
    print("# across_vm = masks held in an SGPR PAIR (s_and_b64 / s_or_b64 / v_cmp .. s[a:b] results) that are read after an intervening s_waitcnt vmcnt:")
    print("#             the pattern of the round-2/3 contention bug.  vcc_vm = the same for VCC (v_cmp -> v_cndmask with a wait scheduled between).")
    print("# mask_ops = s_and/or/xor/andn2.._b64 on non-exec pairs (mask arithmetic); cnd_sgpr = v_cndmask selects driven by an SGPR-pair mask.")
    print(f"# {'file':16s} {'kernel':88s} {'instr':>6s} {'masks':>6s} {'across_vm':>9s} {'max_dist':>8s} {'vcc_vm':>6s} {'exec_saves':>10s} {'mask_ops':>8s} {'cnd_sgpr':>8s}")
    for r in sorted(rows, key=lambda r: (-r[4], -(r[8] + r[9]), r[0], r[1])):
        print(f"{r[0]:18s} {r[1][:88]:88s} {r[2]:6d} {r[3]:6d} {r[4]:9d} {r[5]:8d} {r[7]:6d} {r[6]:10d} {r[8]:8d} {r[9]:8d}")


if __name__ == "__main__":
    main()
