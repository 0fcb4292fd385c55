#!/bin/bash
# Round 5, GPU call 3: Infinity-Cache experiment -- the backbone's front (stem + layer1 + layer2) per group of G images (engine opt bb_group).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r05c3
for dt in bf16 f32s; do
  for g in 0 16 8 4; do
    timeout 200 python bench.py --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-other-dtypes --no-bs1 --engine-opt bb_group=$g > ${O}_${dt}_g$g.json 2>/dev/null
    python - <<P
import json
d=json.loads(open('${O}_${dt}_g$g.json').read().strip().splitlines()[-1])
print('$dt bb_group=$g', d['value'], d['ms_per_step'], d['block_ms'][:4])
P
  done
done
for g in 0 8; do
  timeout 200 python tools/profile_stages.py --dtype bf16 --engine-opt bb_group=$g 2>/dev/null | tail -1
done
