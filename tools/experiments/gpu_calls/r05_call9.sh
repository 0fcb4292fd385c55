#!/bin/bash
# Round 5, last GPU call: the full GPU suite, smoke and the driver's bench command at the final commit (per-symbol PMC traffic attached).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05_pytest_gpu_v4.txt 2>&1; tail -3 gpurun_out/r05_pytest_gpu_v4.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_RC=0')" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r05_bench_v4.json 2> gpurun_out/r05_bench_v4.err; tail -c 400 gpurun_out/r05_bench_v4.json; echo
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05_bench_v4.json').read().strip().splitlines()[-1])
print('value', d['value'], d['ms_per_step'], 'roofline', {k: d['roofline'].get(k) for k in ('symbol','achieved','frac','traffic','mean_launch_ms')}, d['traffic_source'])
for k,v in d['by_dtype'].items():
    p=v.get('parity_vs_oracle') or {}; f=v.get('free_running_v4') or {}
    print(k, v.get('lines_per_s'), 'gate', p.get('parity_gate'), (p.get('teacher_forced') or {}).get('logit_err_max'), (p.get('teacher_forced') or {}).get('strings_identical_same_selection'), 'v4', (f.get('free_running') or {}).get('strings_identical_free_running'), (f.get('free_running') or {}).get('cer_free_running'))
P
