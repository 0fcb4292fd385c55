#!/bin/bash
# Round 6, GPU call 4: the side-stream schedule -- equivalence tests, HIP-graph capture with side streams, same-box A/B of the step
# (overlap_streams = 1 / 0) for bf16 and f32s.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r06c4
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "side_stream or hip_graph or multi_slice or bench_accepts or two_ranks or tiny_model" 2>&1 | tail -4 | cut -c1-600
for dt in bf16 f32s; do
  for ov in 1 0 1 0; do
    timeout 300 python bench.py --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-other-dtypes --no-bs1 --engine-opt overlap_streams=$ov 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$dt overlap=$ov', d['value'], d['ms_per_step'])"
  done
done
