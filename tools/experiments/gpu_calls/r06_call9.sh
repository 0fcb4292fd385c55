#!/bin/bash
# Round 6, GPU call 9: ffn4 variants (slot 0 wrong, slot 1 right in the first build): ramp-in loop on the two-slot body / hazard pads / 4 fragment registers
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 600 python tools/experiments/ffn4_variants.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06c9_ffn4_variants.txt
