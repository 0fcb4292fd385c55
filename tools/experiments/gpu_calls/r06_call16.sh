#!/bin/bash
# Round 6, GPU call 16: encoder sampler, fourth form (reads one half-point ahead, 384 threads) against the third: tests with the experiment
# build, per-operator time, bench step A/B (separate processes: the variant is read once per process)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export DTLR_HIP_LIB=$PWD/dtlr_amd/libdtlr_hip_instr.so
DTLR_MSDA_ENC_V=4 timeout 900 python -m pytest tests/test_gpu_msda.py -q -x -k "encoder" 2>&1 | tail -4 | cut -c1-300
for v in 3 4 3 4; do
  DTLR_MSDA_ENC_V=$v timeout 200 python tools/profile_ops.py --dtype bf16 --steps 3 --top 70 2>/dev/null | grep "msda_encoder\|step_ms" | head -2 | sed "s/^/V=$v  /"
done | tee gpurun_out/r06c16_msda_v4.txt
for v in 3 4 3 4; do
  DTLR_MSDA_ENC_V=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-other-dtypes --no-bs1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 DTLR_MSDA_ENC_V=$v', d['value'], d['ms_per_step'], d['roofline_msda']['mean_launch_ms'])"
done | tee -a gpurun_out/r06c16_msda_v4.txt
