#!/bin/bash
# Round-end verification on the GPU box (run through gpurun): GPU tests, smoke, bench (the driver's command), stage / operator splits, rocprofv3 kernel
# stats, secondary configurations.  Outputs land in gpurun_out/ (copy what should be judged into profiles/).
# usage: [BENCH_EXTRA="--no-cpu-baseline"] [ROUND=r04] bash tools/final_check.sh <tag> [prof-only|no-tests]
TAG=${1:-vX}
R4=${ROUND:-r06}
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
if [ "$2" != "prof-only" ]; then
  if [ "$2" != "no-tests" ]; then
    timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${R4}_pytest_gpu_$TAG.txt 2>&1; tail -4 gpurun_out/${R4}_pytest_gpu_$TAG.txt | cut -c1-300
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_RC=0')" 2>&1 | tail -2
  fi
  timeout 900 python bench.py --steps 20 --warmup 3 $BENCH_EXTRA > gpurun_out/${R4}_bench_$TAG.json 2> gpurun_out/${R4}_bench_$TAG.err; tail -c 300 gpurun_out/${R4}_bench_$TAG.json; echo
  for dt in bf16 f32s; do
    timeout 200 python tools/profile_stages.py --dtype $dt > gpurun_out/${R4}_stage_split_${dt}_$TAG.json 2>/dev/null; tail -1 gpurun_out/${R4}_stage_split_${dt}_$TAG.json
    timeout 200 python tools/profile_ops.py --dtype $dt --top 60 > gpurun_out/${R4}_ops_by_shape_${dt}_$TAG.txt 2>/dev/null; head -4 gpurun_out/${R4}_ops_by_shape_${dt}_$TAG.txt
  done
  for cfgx in "chinese:--config chinese" "latin_mixed:--config latin-mixed" "latin_eval:--config latin-eval" "swinT:--backbone swin_T_224_1k"; do
    nm=${cfgx%%:*}; fl=${cfgx#*:}
    timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-dtypes $fl > gpurun_out/${R4}_bench_${nm}_$TAG.json 2>/dev/null
    python - <<P
import json
try:
    d=json.loads(open('gpurun_out/${R4}_bench_${nm}_$TAG.json').read().strip().splitlines()[-1])
    c=d.get('parity_vs_oracle') or {}
    print('$nm', d['value'], d['ms_per_step'], c.get('teacher_forced'), (c.get('free_running') or {}).get('strings_identical_free_running'))
except Exception as e: print('$nm', 'failed', e)
P
  done
fi
export TMPDIR=/tmp
R=$PWD
for dt in bf16 f32s; do
  mkdir -p gpurun_out/prof_${dt}_$TAG
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${dt}_$TAG -o r -- python $R/bench.py --dtype $dt --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-other-dtypes --no-bs1 > $R/gpurun_out/${R4}_bench_prof_${dt}_$TAG.json 2>/dev/null )
  python tools/rocprof_summary.py gpurun_out/prof_${dt}_$TAG gpurun_out/${R4}_bench_${dt}_b32_$TAG
  head -8 gpurun_out/${R4}_bench_${dt}_b32_${TAG}_kernel_stats.csv | cut -c1-200
  rm -rf gpurun_out/prof_${dt}_$TAG
done
