#!/bin/bash
# Round 5, GPU call 2: generator v4 (rank-invariant, image-driven characters): free-running strings of all four engines against the oracle;
# the per-stream split-K workspace under two overlapped streams; the bench line with the v4 leg.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r05c2
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "splitk_convolutions_overlapped or conv2d_nhwc" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -q -s -k "free_running_strings_vs_oracle_on_image_driven" 2>&1 | grep -E "passed|failed|^E  |Error|free-running on v4" | cut -c1-1500 | tail -12
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > ${O}_bench.json 2> ${O}_bench.err; tail -c 300 ${O}_bench.json; echo
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05c2_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], d['ms_per_step'])
for k,v in d['by_dtype'].items():
    p=v.get('parity_vs_oracle') or {}; f=v.get('free_running_v4') or {}
    print(k, v.get('lines_per_s'), 'gate', p.get('parity_gate'), 'v2 TF', {x:(p.get('teacher_forced') or {}).get(x) for x in ('logit_err_max','strings_identical_same_selection')},
          'V4 FREE', {x:(f.get('free_running') or {}).get(x) for x in ('strings_identical_free_running','cer_free_running','rank_slots_changed','two_stage_score_err_max','oracle_vs_itself_at_that_score_error')},
          'V4 TF', {x:(f.get('teacher_forced') or {}).get(x) for x in ('logit_err_max','cer_same_selection','label_flips','min_oracle_margin_on_flipped')}, f.get('error'))
P
grep -E "oracle free run|Traceback|Error|timed" ${O}_bench.err | head
