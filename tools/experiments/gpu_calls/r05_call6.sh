#!/bin/bash
# Round 5, GPU call 6: ffn_split's hidden-dimension split of the last partial round (tests, per-operator times, f32s step); graph-replay
# regression test; bench on user assets.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r05c6
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -s -k "ffn_split or splitk or rowmax" 2>&1 | grep -E "passed|failed|^E  |Error|ffn_split M" | tail -20
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "graph_replay or user_checkpoint or (f32s and (bench_batch or full_size or tiny))" 2>&1 | tail -4
timeout 200 python tools/profile_ops.py --dtype f32s --steps 3 --top 12 2>/dev/null | head -8 | cut -c1-150
timeout 300 python bench.py --dtype f32s --steps 10 --warmup 2 --no-cpu-baseline --no-other-dtypes --no-bs1 --parity-lines 4 > ${O}_bench_f32s.json 2>/dev/null
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05c6_bench_f32s.json').read().strip().splitlines()[-1])
p=d.get('parity_vs_oracle') or {}
print('f32s', d['value'], d['ms_per_step'], p.get('parity_gate'), (p.get('teacher_forced') or {}).get('logit_err_max'), ((d.get('free_running_v4') or {}).get('free_running') or {}).get('strings_identical_free_running'))
P
