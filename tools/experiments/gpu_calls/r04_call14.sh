#!/bin/bash
# Round 4, GPU call 14: fp32 encoder sampler with the query-slot piece rotation (bank conflicts): MSDA tests, f32 / f32s parity, f32s step.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04c14
timeout 400 python -m pytest tests/test_gpu_msda.py -q -x 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -k "f32s or float32 or f32" -s 2>&1 | grep -E "passed|failed|^E  |Error" | cut -c1-260 | tail -6
timeout 200 python tools/profile_ops.py --dtype f32s --steps 2 --top 8 2>/dev/null | head -6
timeout 300 python bench.py --dtype f32s --steps 10 --warmup 2 --no-cpu-baseline --no-other-dtypes --no-bs1 > ${O}_bench_f32s.json 2> ${O}_bench_f32s.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04c14_bench_f32s.json').read().strip().splitlines()[-1])
print('f32s', d['value'], d['ms_per_step'], {k:d['cer_vs_oracle'].get(k) for k in ('logit_err_max','cx_err_max','cer_all_queries','unexplained')})
P
