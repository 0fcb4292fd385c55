#!/bin/bash
# Round 6, GPU call 3: full GPU suite + smoke + the driver's bench line + per-operator times, at the kernels of the first half of the round
# (split conv3x3 patch kernel, six-slice decoder value projection, workspace lifetime, inf flag).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r06c3
timeout 1700 python -m pytest tests -m gpu -q > ${O}_pytest_gpu.txt 2>&1; tail -4 ${O}_pytest_gpu.txt | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_RC=0')" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > ${O}_bench.json 2> ${O}_bench.err
echo "stdout bytes: $(wc -c < ${O}_bench.json)"; python - <<'P'
import json
d=json.loads(open('gpurun_out/r06c3_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v.get('lines_per_s'), v.get('strings_teacher_forced'), v.get('strings_free_running_v4')) for k,v in d['by_dtype'].items()})
P
cp gpurun_out/bench_detail.json ${O}_bench_detail.json 2>/dev/null
for dt in f32s bf16; do
  timeout 200 python tools/profile_ops.py --dtype $dt --steps 3 --top 70 > ${O}_ops_${dt}.txt 2>/dev/null; head -2 ${O}_ops_${dt}.txt | cut -c1-150
  timeout 200 python tools/profile_stages.py --dtype $dt > ${O}_stage_${dt}.json 2>/dev/null; tail -1 ${O}_stage_${dt}.json
done
