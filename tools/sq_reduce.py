#!/usr/bin/env python3
"""Per-kernel ratios from one `rocprofv3 --pmc SQ_*` pass (tools/sq_pass.sh): sums over all dispatches and all counter instances.
    python tools/sq_reduce.py <prefix>_counters.csv <prefix>_kernel_stats.csv
Columns (ratios of SUMS, so the unit of a cycle counter cancels as long as both counters tick in the same unit -- the SQ counts
"busy" counters in quad-cycles on gfx9):
  mfma/cu     SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): matrix-pipe busy share of the busy CU time (4 SIMDs per CU)
  lds/cu      SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES: share of busy CU cycles in which the LDS index unit is active
  conflict    SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: share of the LDS-active cycles lost to bank conflicts
  valu/wave   SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, ldsi/wave SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES: issue activity per resident wave-cycle"""
import csv
import re
import sys


def main():
    ctr, stats = sys.argv[1], sys.argv[2]
    c = {}
    for row in csv.reader(l for l in open(ctr) if not l.startswith("#")):
        if len(row) < 5 or row[0] == "kernel":
            continue
        c.setdefault(row[0], {})[row[1]] = float(row[3])
    rows = [r for r in csv.reader(l for l in open(stats) if not l.startswith("#")) if len(r) >= 5 and r[0] != "name"]
    print(f"{'ms total':>9} {'calls':>6} {'avg us':>8} {'mfma/cu':>8} {'lds/cu':>7} {'conflict':>8} {'valu/wave':>9} {'ldsi/wave':>9}  kernel")
    for r in rows[:24]:
        k = c.get(r[0])
        if not k:
            continue
        cu, wv = k.get("SQ_BUSY_CU_CYCLES", 0.0), k.get("SQ_WAVE_CYCLES", 0.0)
        idx = k.get("SQ_LDS_IDX_ACTIVE", 0.0)

        def q(a, b):
            return f"{a / b:.3f}" if b else "-"
        name = re.sub(r"\(.*", "", r[0]).replace("void ", "").replace("dtlr::", "")
        print(f"{float(r[2]):9.2f} {r[1]:>6} {float(r[4]):8.1f} {q(k.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), 4 * cu):>8} {q(idx, cu):>7} "
              f"{q(k.get('SQ_LDS_BANK_CONFLICT', 0.0), idx):>8} {q(k.get('SQ_ACTIVE_INST_VALU', 0.0), wv):>9} {q(k.get('SQ_ACTIVE_INST_LDS', 0.0), wv):>9}  {name[:90]}")


if __name__ == "__main__":
    main()
