#!/bin/bash
# Round 4, GPU call 11: split stem + weight-resident K = 256 split projection (kernel tests, engine parity, f32s profile).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04c11
{
echo "== kernel tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "stem_conv7x7_split or gemm_k256s" -s 2>&1 | grep -E "split stem|k256s |passed|failed|^E  " | cut -c1-220 | tail -40
echo "== engine parity (f32s)"; timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "f32s" -s 2>&1 | grep -E "vs reference|passed|failed|^E  |Error" | cut -c1-260 | tail -30
} > ${O}_tests.txt 2>&1
tail -40 ${O}_tests.txt
timeout 300 python tools/profile_stages.py --dtype f32s --steps 3 > ${O}_stage_f32s.json 2>/dev/null; tail -1 ${O}_stage_f32s.json
timeout 300 python tools/profile_ops.py --dtype f32s --steps 2 --top 40 > ${O}_ops_f32s.txt 2>/dev/null; head -20 ${O}_ops_f32s.txt
timeout 300 python bench.py --dtype f32s --steps 10 --warmup 2 --no-cpu-baseline --no-other-dtypes > ${O}_bench_f32s.json 2> ${O}_bench_f32s.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04c11_bench_f32s.json').read().strip().splitlines()[-1])
print('f32s', d['value'], d['ms_per_step'], d.get('latency_ms_by_batch'), {k:d['cer_vs_oracle'].get(k) for k in ('logit_err_max','cx_err_max','cer_all_queries','unexplained')})
for r in d['roofline_by_kernel']: print(r['kernel'][:40], {k:r[k] for k in ('bound','achieved','frac','mean_launch_ms','ms_per_step') if k in r})
P
