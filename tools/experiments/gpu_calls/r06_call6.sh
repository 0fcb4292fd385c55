#!/bin/bash
# Round 6, GPU call 6 (second session): full GPU suite at HEAD (call 3 had one stale assertion in the --weights/--images test), smoke, the driver's
# bench command, and the VALU issue-rate microbenchmark behind the encoder sampler's cost model (tools/experiments/valu_rate.hip).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r06c6
timeout 120 tools/experiments/valu_rate > ${O}_valu_rate.txt 2>&1; cat ${O}_valu_rate.txt
timeout 1700 python -m pytest tests -m gpu -q > ${O}_pytest_gpu.txt 2>&1; tail -4 ${O}_pytest_gpu.txt | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_RC=0')" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > ${O}_bench.json 2> ${O}_bench.err
echo "stdout bytes: $(wc -c < ${O}_bench.json)"; python - <<'P'
import json
d=json.loads(open('gpurun_out/r06c6_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v.get('lines_per_s'), v.get('strings_teacher_forced'), v.get('strings_free_running_v4')) for k,v in d['by_dtype'].items()})
P
cp gpurun_out/bench_detail.json ${O}_bench_detail.json 2>/dev/null
