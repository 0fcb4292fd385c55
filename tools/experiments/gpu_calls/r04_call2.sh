#!/bin/bash
# Round 4, GPU call 2: fp32 level-per-lane MSDA encoder form, split attention, specialised split epilogues, promoted 16-bit defaults.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04c2
{
echo "== kernel tests (msda encoder, mha, split)"; timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "split or mha or msda_encoder or msda_far or tall_canvas" -s 2>&1 | grep -E "split gemm|split mha|passed|failed|^E  " | cut -c1-220 | tail -40
echo "== engine parity (f32, f32s)"; timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "f32 or tiny" -s 2>&1 | grep -E "vs reference|passed|failed|^E  |Error" | cut -c1-240 | tail -30
} > ${O}_tests.txt 2>&1
tail -30 ${O}_tests.txt
for dt in f32s; do
  timeout 300 python tools/profile_stages.py --dtype $dt --steps 3 > ${O}_stage_$dt.json 2>/dev/null; tail -1 ${O}_stage_$dt.json
  timeout 300 python tools/profile_ops.py --dtype $dt --steps 2 --top 45 > ${O}_ops_$dt.txt 2>/dev/null; head -24 ${O}_ops_$dt.txt
done
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > ${O}_bench.json 2> ${O}_bench.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r04c2_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('latency_ms_by_batch'))
for k,v in d['by_dtype'].items():
    c=v.get('cer_vs_oracle') or {}
    print(k, v.get('lines_per_s'), v.get('ms_per_step'), {x:c.get(x) for x in ('logit_err_max','cx_err_max','cer_all_queries','label_flips','order_swaps','unexplained')}, v.get('error'))
P
