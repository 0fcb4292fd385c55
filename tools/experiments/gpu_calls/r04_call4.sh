#!/bin/bash
# Round 4, GPU call 4: where does the f32s (and f32?) engine differ between forwards -- alone, and with two processes on the GPU?
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for k in f32s f32; do
  timeout 300 python tools/experiments/contention_probe2.py 30 solo_$k $k > gpurun_out/r04c4_solo_$k.log 2>&1
  grep -h "done\|forward\"" gpurun_out/r04c4_solo_$k.log | cut -c1-1500
  timeout 400 python tools/experiments/contention_probe2.py 40 A_$k $k > gpurun_out/r04c4_A_$k.log 2>&1 &
  PA=$!
  timeout 400 python tools/experiments/contention_probe2.py 40 B_$k $k > gpurun_out/r04c4_B_$k.log 2>&1 &
  PB=$!
  wait $PA $PB
  grep -h "done\|forward\"\|Error" gpurun_out/r04c4_A_$k.log gpurun_out/r04c4_B_$k.log | cut -c1-1500
done
timeout 900 python -m pytest tests/test_gpu_model.py -q -k "two_processes and not f32" 2>&1 | tail -4
