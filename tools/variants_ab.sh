#!/bin/bash
# First GPU call of a round that starts with banked (compiled, non-default, never run) kernel variants: test them, then time each one
# against the default in the SAME call (same box, same clocks).   usage (through gpurun): bash tools/variants_ab.sh [tag]
# Outputs: gpurun_out/variants_<tag>.txt (test verdicts + one "engine_opts lines/s ms" row per run).
TAG=${1:-v1}
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
OUT=gpurun_out/variants_$TAG.txt
: > $OUT
DTLR_TEST_UNTIMED_VARIANTS=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "variant3 or two_pass" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-220 | head -20 | tee -a $OUT
run() {
  timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-other-dtypes "$@" 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['engine_opts'], d['value'], d['ms_per_step'])" | tee -a $OUT
}
run
run --lib-variant msda_enc=3
run --lib-variant mha=1
run
# parity of the combination on 8 lines of the bench batch (cer_vs_oracle of the bf16 line)
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-dtypes --lib-variant msda_enc=3 --lib-variant mha=1 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cer_vs_oracle with both variants:', d['cer_vs_oracle'])" | tee -a $OUT
