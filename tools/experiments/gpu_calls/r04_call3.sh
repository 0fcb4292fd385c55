#!/bin/bash
# Round 4, GPU call 3: the whole GPU suite at the current HEAD + the f32s engine with the 256-row tiles / broadcast residual.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04c3
timeout 1500 python -m pytest tests -m gpu -q -x > ${O}_pytest_gpu.txt 2>&1; tail -5 ${O}_pytest_gpu.txt | cut -c1-300
timeout 300 python tools/profile_stages.py --dtype f32s --steps 3 > ${O}_stage_f32s.json 2>/dev/null; tail -1 ${O}_stage_f32s.json
timeout 300 python tools/profile_ops.py --dtype f32s --steps 2 --top 40 > ${O}_ops_f32s.txt 2>/dev/null; head -22 ${O}_ops_f32s.txt
timeout 300 python bench.py --dtype f32s --steps 10 --warmup 2 --no-cpu-baseline --no-other-dtypes --no-parity > ${O}_bench_f32s.json 2> ${O}_bench_f32s.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04c3_bench_f32s.json').read().strip().splitlines()[-1])
print('f32s', d['value'], d['ms_per_step'], d.get('latency_ms_by_batch'))
for g in d['gemm_by_shape'][:12]: print(g['shape'], g['mean_launch_us'], g['ms_per_step'], g['mfma_frac'], g['hbm_frac'])
P
