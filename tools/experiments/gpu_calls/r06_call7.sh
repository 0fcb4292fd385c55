#!/bin/bash
# Round 6, GPU call 7: first run of the persistent two-slot fused FFN (dtlr_ffn4_bf16): correctness against the reference and ffn32, timing.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "ffn4" 2>&1 | tail -15 | cut -c1-300
timeout 300 python tools/experiments/ffn4_bench.py 2>&1 | tail -8 | tee gpurun_out/r06c7_ffn4_bench.txt
