#!/bin/bash
# Round 6, GPU call 14: ffn4 -- period time with nothing but chunk steps (no epilogue, no loads, no seeding)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for lag in 2 258 770; do FFN4_M=65536,393216 DTLR_FFN4_LAG=$lag timeout 300 python tools/experiments/ffn4_scaling.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06c14_ffn4_bare.txt
