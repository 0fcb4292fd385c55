#!/bin/bash
# Round-end verification on the GPU box (run through gpurun): GPU tests, smoke, bench, stage/operator splits, rocprofv3 kernel stats.
# Outputs land in gpurun_out/ (copy what should be judged into profiles/).   usage: [BENCH_EXTRA="--no-cpu-baseline"] bash tools/final_check.sh <tag> [prof-only]
TAG=${1:-vX}
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
if [ "$2" != "prof-only" ]; then
  timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_$TAG.txt 2>&1; tail -3 gpurun_out/pytest_gpu_$TAG.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_RC=0')" 2>&1 | tail -2
  timeout 900 python bench.py --steps 20 --warmup 3 $BENCH_EXTRA > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -c 400 gpurun_out/bench_$TAG.json; echo
  timeout 200 python tools/profile_stages.py > gpurun_out/stage_$TAG.json 2>/dev/null; tail -1 gpurun_out/stage_$TAG.json
  timeout 200 python tools/profile_ops.py --top 60 > gpurun_out/ops_$TAG.txt 2>/dev/null; head -5 gpurun_out/ops_$TAG.txt
fi
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/prof_$TAG
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-other-dtypes > $R/gpurun_out/bench_prof_$TAG.json 2>/dev/null )
python tools/rocprof_summary.py gpurun_out/prof_$TAG gpurun_out/r03_bench_bf16_b32_$TAG
head -12 gpurun_out/r03_bench_bf16_b32_${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG
