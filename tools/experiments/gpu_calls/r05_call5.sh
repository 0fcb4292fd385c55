#!/bin/bash
# Round 5, GPU call 5: graph replay probe after the row-max initialisation became a kernel; rowmax tests; bench on user assets (traceback).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for dt in f32 bf16; do echo "== graph probe $dt"; timeout 300 python tools/experiments/graph_replay_probe2.py $dt 2>&1 | grep -vE "amdgpu.ids" | tail -8; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "rowmax" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "user_checkpoint" 2>&1 | grep -vE "amdgpu.ids" | tail -40
