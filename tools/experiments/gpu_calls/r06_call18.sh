#!/bin/bash
# Round 6, GPU call 18: HIP-graph replay of the forward + decode against eager launches at 1, 4 and 32 lines (the single-line latency case)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for b in 1 1 4 32; do timeout 300 python tools/try_graph.py $b 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/B=$b /"; done | tee gpurun_out/r06c18_graph_bs.txt
