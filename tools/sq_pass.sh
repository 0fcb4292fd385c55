#!/bin/bash
# SQ counter pass on the GPU box (run through gpurun): matrix-pipe busy cycles, LDS activity and bank conflicts, VALU activity per kernel of a
# bench step, bf16 and f32s engines; rocprofv3 --kernel-trace --pmc only (one pass per engine).   usage: bash tools/sq_pass.sh <tag>
TAG=${1:-vX}
RN=${ROUND:-r06}
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
for dt in bf16 f32s; do
  D=$R/gpurun_out/sq_${dt}_$TAG
  mkdir -p $D
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $D -o sq -- python $R/bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-other-dtypes --no-bs1 2>&1 | tail -1 | cut -c1-200 )
  python tools/rocprof_summary.py $D gpurun_out/${RN}_sq_${dt}_$TAG > /dev/null 2>&1
  python tools/sq_reduce.py gpurun_out/${RN}_sq_${dt}_${TAG}_counters.csv gpurun_out/${RN}_sq_${dt}_${TAG}_kernel_stats.csv > gpurun_out/${RN}_sq_${dt}_$TAG.txt
  head -14 gpurun_out/${RN}_sq_${dt}_$TAG.txt | cut -c1-200
  rm -rf $D
done
