#!/bin/bash
# Round 6, GPU call 1: the driver's bench command -> ONE stdout line below 6 KB (compact_line) + bench_detail.json; baseline per-operator
# times of the bf16 and f32s steps at this round's starting kernels.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r06c1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > ${O}_bench.json 2> ${O}_bench.err
echo "stdout bytes: $(wc -c < ${O}_bench.json) lines: $(wc -l < ${O}_bench.json)"; cat ${O}_bench.json
cp gpurun_out/bench_detail.json ${O}_bench_detail.json 2>/dev/null
for dt in f32s bf16; do
  timeout 200 python tools/profile_ops.py --dtype $dt --steps 3 --top 70 > ${O}_ops_${dt}.txt 2>/dev/null; head -3 ${O}_ops_${dt}.txt | cut -c1-150
  timeout 200 python tools/profile_stages.py --dtype $dt > ${O}_stage_${dt}.json 2>/dev/null; tail -1 ${O}_stage_${dt}.json
done
