#!/bin/bash
# Round 6, GPU call 8: error map of dtlr_ffn4_bf16 (first run: ~0.02 off the reference)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 300 python tools/experiments/ffn4_debug.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06c8_ffn4_debug.txt | cut -c1-220
