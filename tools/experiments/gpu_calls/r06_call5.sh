#!/bin/bash
# Round 6, GPU call 5: small f32s decoder projections on the streaming kernel -- model test + same-box A/B of the f32s step.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -s -k "multi_slice or tiny_model or side_stream" 2>&1 | grep -E "passed|failed|^E  |k256s_multi in" | cut -c1-300 | tail -8
for ov in 1 0 1 0; do
  timeout 300 python bench.py --dtype f32s --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-other-dtypes --no-bs1 --engine-opt use_k256s_small=$ov 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32s use_k256s_small=$ov', d['value'], d['ms_per_step'])"
done
