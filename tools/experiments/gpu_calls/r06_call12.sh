#!/bin/bash
# Round 6, GPU call 12: ffn4 -- what do the epilogue steps cost?  time against rows with and without the LayerNorm / store slices
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for lag in 2 258; do DTLR_FFN4_LAG=$lag timeout 300 python tools/experiments/ffn4_scaling.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06c12_ffn4_noepi.txt
