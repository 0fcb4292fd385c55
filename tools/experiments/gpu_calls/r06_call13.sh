#!/bin/bash
# Round 6, GPU call 13: ffn4 -- weight-fragment prefetch depth 2 / 3 / 4 groups (time per tile-pair period, with and without epilogue slices)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for lib in libffn4_lagrt.so libffn4_nw3.so libffn4_nw4.so; do for lag in 2 258; do echo "== $lib lag $lag"; FFN4_LIB=$lib FFN4_M=65536,174080,393216 DTLR_FFN4_LAG=$lag timeout 300 python tools/experiments/ffn4_scaling.py 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r06c13_ffn4_nw.txt
