#!/bin/bash
# Round 5, GPU call 1: the round's parity legs (free-running + teacher-forced, oracle/parity.py) on hardware, the split GEMM's new LDS row
# swizzle (tests + per-operator times of the f32s step), the bench line with the per-kernel roofline.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r05c1
timeout 500 python -m pytest tests/test_gpu_kernels.py -q -x -k "split or f32s" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -s -k "parity_engines_bench or full_model_vs_golden or tiny_model" 2>&1 | grep -E "passed|failed|^E  |Error|free-running|bs32 vs oracle" | cut -c1-1800 | tail -14
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 200 python tools/profile_ops.py --dtype f32s --steps 3 --top 40 > ${O}_ops_f32s.txt 2>/dev/null; head -24 ${O}_ops_f32s.txt | cut -c1-150
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > ${O}_bench.json 2> ${O}_bench.err; tail -c 600 ${O}_bench.json; echo
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05c1_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], d['ms_per_step'], 'roofline', {k: d['roofline'].get(k) for k in ('symbol','bound','achieved','frac','mean_launch_ms')})
for k,v in d['by_dtype'].items():
    p=v.get('parity_vs_oracle') or {}
    print(k, v.get('lines_per_s'), v.get('ms_per_step'), 'gate', p.get('parity_gate'), 'TF', {x:(p.get('teacher_forced') or {}).get(x) for x in ('logit_err_max','cx_err_max','strings_identical_same_selection','cer_same_selection','within_budget')},
          'FREE', {x:(p.get('free_running') or {}).get(x) for x in ('strings_identical_free_running','cer_free_running','lines_with_identical_selection','strings_identical_given_identical_selection','rank_slots_changed','two_stage_score_err_max','oracle_vs_itself_at_that_score_error')})
P
grep -E "oracle free run|Traceback|Error" ${O}_bench.err | head
