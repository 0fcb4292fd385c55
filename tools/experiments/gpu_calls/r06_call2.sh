#!/bin/bash
# Round 6, GPU call 2: the multi-slice split projection kernel and the split conv3x3 patch kernel: unit tests, the round's other new GPU
# tests (workspace lifetime, inf flag), the f32s engine's model tests, per-operator times of the f32s step.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r06c2
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -s -k "k256s_multi or conv3x3_patch_f32s or workspace_growth or non_finite" 2>&1 | grep -E "passed|failed|^E  |Error|max err" | cut -c1-400 | tail -30
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "f32s or split or parity_engines or free_running or hip_graph" 2>&1 | tail -5 | cut -c1-600
timeout 200 python tools/profile_ops.py --dtype f32s --steps 3 --top 70 > ${O}_ops_f32s.txt 2>/dev/null; head -30 ${O}_ops_f32s.txt | cut -c1-150
timeout 200 python tools/profile_stages.py --dtype f32s > ${O}_stage_f32s.json 2>/dev/null; tail -1 ${O}_stage_f32s.json
