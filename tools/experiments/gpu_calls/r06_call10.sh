#!/bin/bash
# Round 6, GPU call 10: ffn4 -- lag of slot 1 behind slot 0 (runtime argument), correctness + time of the encoder call per lag
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for lag in 1 2 3 4 6 33; do
  echo "== lag $lag"
  DTLR_FFN4_LAG=$lag timeout 300 python tools/experiments/ffn4_variants.py 2>&1 | grep -v amdgpu.ids | grep -v "M=   256\|M=  1024"
done | tee gpurun_out/r06c10_ffn4_lag.txt
