#!/bin/bash
# Round 6, GPU call 11: ffn4 vs ffn32, time against rows (per-period time vs fixed part), lag 2 and 33
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for lag in 2 33; do DTLR_FFN4_LAG=$lag timeout 300 python tools/experiments/ffn4_scaling.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06c11_ffn4_scaling.txt
