#!/bin/bash
# SQ counters of the two encoder-FFN kernels (separate passes; --kernel-trace only).  usage: bash tools/experiments/ffn32_pmc.sh
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
export TMPDIR=/tmp
i=0
for C in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  D=$R/gpurun_out/ffn32_pmc_$i
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $C -d $D -o pmc -- python $R/tools/experiments/ffn32_bench.py > /dev/null 2>&1 )
  python tools/rocprof_summary.py $D gpurun_out/ffn32_pmc_$i > /dev/null 2>&1
  grep -i "ffn" gpurun_out/ffn32_pmc_${i}_counters.csv | sed -e 's/"void dtlr::\([a-z0-9_]*\)[^"]*"/\1/' | cut -c1-120
  rm -rf $D
done
grep -i "ffn" gpurun_out/ffn32_pmc_1_kernel_stats.csv | sed -e 's/"void dtlr::\([a-z0-9_<>]*\)[^"]*"/\1/' | cut -c1-120
