#!/bin/bash
# Round 4, GPU call 13: k256s with the early DMA issue, compile-time mask, LN-only form (two-stage front end): kernel tests, microbench, parity, step.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04c13
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm_k256s" 2>&1 | tail -2
timeout 100 python tools/experiments/k256s_bench.py
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "f32s" -s 2>&1 | grep -E "vs reference|passed|failed|^E  |Error" | cut -c1-260 | tail -12
timeout 300 python tools/profile_stages.py --dtype f32s --steps 3 2>/dev/null | tail -1
timeout 300 python bench.py --dtype f32s --steps 10 --warmup 2 --no-cpu-baseline --no-other-dtypes --no-bs1 > ${O}_bench_f32s.json 2> ${O}_bench_f32s.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04c13_bench_f32s.json').read().strip().splitlines()[-1])
print('f32s', d['value'], d['ms_per_step'], {k:d['cer_vs_oracle'].get(k) for k in ('logit_err_max','cx_err_max','cer_all_queries','unexplained')})
P
