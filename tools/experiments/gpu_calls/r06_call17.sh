#!/bin/bash
# Round 6, GPU call 17: the bench line with `parity_grade` as the driver runs it + the tests that run bench.py as a subprocess
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "bench" 2>&1 | tail -3 | cut -c1-300
timeout 900 python bench.py > gpurun_out/r06_bench_v3.json 2> gpurun_out/r06_bench_v3.err
echo "stdout bytes: $(wc -c < gpurun_out/r06_bench_v3.json), lines: $(wc -l < gpurun_out/r06_bench_v3.json)"; python - <<'P'
import json
d=json.loads(open('gpurun_out/r06_bench_v3.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('parity_grade'), d['steps'], d['warmup'])
P
