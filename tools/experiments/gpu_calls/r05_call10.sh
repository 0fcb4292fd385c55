#!/bin/bash
# Round 5, GPU call 10: the token-stationary class head (dtlr_head_ts): kernel tests, the Chinese model's heads against the tiled forms,
# the Chinese step with / without it, the Latin step with the two-stage scores on it (A/B through the engine attribute).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r05c10
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "head_ts" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -s -k "chinese" 2>&1 | grep -E "passed|failed|^E  |Error|token-stationary" | cut -c1-300 | tail -8
for mc in 1024 1000000; do
  timeout 400 python bench.py --config chinese --steps 10 --warmup 2 --no-cpu-baseline --no-other-dtypes --no-bs1 --parity-lines 4 --engine-opt head_ts_min_classes=$mc > ${O}_chinese_mc$mc.json 2>/dev/null
  python - <<P
import json
d=json.loads(open('${O}_chinese_mc$mc.json').read().strip().splitlines()[-1])
p=d.get('parity_vs_oracle') or {}
print('chinese head_ts_min_classes=$mc', d['value'], d['ms_per_step'], (p.get('teacher_forced') or {}).get('logit_err_max'), p.get('parity_gate'))
for r in d['gemm_by_shape'][:5]: print('   ', r['shape'], r['launches_per_step'], r['mean_launch_us'], r['mfma_frac'])
P
done
for mc in 0 1024; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-dtypes --no-bs1 --no-parity --engine-opt head_ts_min_classes=$mc > ${O}_latin_mc$mc.json 2>/dev/null
  python - <<P
import json
d=json.loads(open('${O}_latin_mc$mc.json').read().strip().splitlines()[-1])
print('latin head_ts_min_classes=$mc', d['value'], d['ms_per_step'])
for r in d['gemm_by_shape']:
    if 'rowmax' in r['shape'] or 'head_ts' in r['shape']: print('   ', r['shape'], r['mean_launch_us'], r['mfma_frac'])
P
done
