#!/bin/bash
# Round 5, GPU call 11: dtlr_head_ts with the LDS-transposed logits stores; scores on it for every charset.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r05c11
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "head_ts or rowmax" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -s -k "chinese_heads or bf16_engine_vs_oracle or tiny_model" 2>&1 | grep -E "passed|failed|^E  |Error|token-stationary" | cut -c1-300 | tail -8
timeout 400 python bench.py --config chinese --steps 10 --warmup 2 --no-cpu-baseline --no-other-dtypes --no-bs1 --parity-lines 4 > ${O}_chinese.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-dtypes --no-bs1 --parity-lines 4 > ${O}_latin.json 2>/dev/null
python - <<'P'
import json
for nm in ('chinese','latin'):
    d=json.loads(open(f'gpurun_out/r05c11_{nm}.json').read().strip().splitlines()[-1])
    p=d.get('parity_vs_oracle') or {}
    print(nm, d['value'], d['ms_per_step'], (p.get('teacher_forced') or {}).get('logit_err_max'), p.get('parity_gate'), ((d.get('free_running_v4') or {}).get('free_running') or {}).get('cer_free_running'))
    for r in d['gemm_by_shape']:
        if 'head_ts' in r['shape'] or 'rowmax' in r['shape']: print('   ', r['shape'], r['mean_launch_us'], r['mfma_frac'], r['hbm_frac'])
P
