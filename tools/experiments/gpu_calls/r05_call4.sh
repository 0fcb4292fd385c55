#!/bin/bash
# Round 5, GPU call 4: where does a HIP-graph replay stop reproducing itself after an eager forward; bench on a user
# checkpoint + image folder (synthetic assets); full GPU suite of the files touched this round.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for dt in bf16 f32; do echo "== graph probe $dt"; timeout 300 python tools/experiments/graph_replay_probe2.py $dt 2>&1 | grep -vE "amdgpu.ids" | tail -12; done
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "user_checkpoint" 2>&1 | tail -4
