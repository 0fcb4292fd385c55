#!/bin/bash
# Round 4, GPU call 1: the split-fp16 (f32s) kernels and engine, the banked round-3 variants, the bench self-launch.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04c1
{
echo "== split kernel tests"; timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -k "split" -s 2>&1 | grep -E "split gemm|passed|failed|^E  " | cut -c1-200 | tail -40
echo "== f32s engine parity"; timeout 900 python -m pytest tests/test_gpu_model.py -q -k "f32s" -s 2>&1 | grep -E "vs reference|passed|failed|^E  |Error" | cut -c1-240 | tail -30
echo "== self-launch"; timeout 600 python -m pytest tests/test_gpu_model.py -q -k "starts_its_own_ranks or two_ranks_on_one_gpu" 2>&1 | tail -3
} > ${O}_tests.txt 2>&1
tail -40 ${O}_tests.txt
for dt in f32s f32; do
  timeout 300 python tools/profile_stages.py --dtype $dt --steps 3 > ${O}_stage_$dt.json 2>/dev/null; tail -1 ${O}_stage_$dt.json
  timeout 300 python tools/profile_ops.py --dtype $dt --steps 2 --top 45 > ${O}_ops_$dt.txt 2>/dev/null; head -30 ${O}_ops_$dt.txt
done
timeout 600 python bench.py --dtype f32s --steps 10 --warmup 2 --no-cpu-baseline --no-other-dtypes > ${O}_bench_f32s.json 2> ${O}_bench_f32s.err; tail -c 1500 ${O}_bench_f32s.json; echo
bash tools/variants_ab.sh r04c1
