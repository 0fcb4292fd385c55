#!/bin/bash
# Round 6, GPU call 15: encoder FFN dispatch A/B (ffn32 rounds + 16x16x32 tail vs ffn32 over all rows) and the same A/B on the whole bench step
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 300 python tools/experiments/ffn_tail_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06c15_ffn_tail_ab.txt
for v in 1 0 1 0; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-other-dtypes --no-bs1 --engine-opt ffn32_tail=$v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 ffn32_tail=$v', d['value'], d['ms_per_step'])"
done | tee -a gpurun_out/r06c15_ffn_tail_ab.txt
