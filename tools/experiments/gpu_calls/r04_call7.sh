#!/bin/bash
# Round 4, GPU call 7: which change to the victim kernel removes the effect (aggressor: gemm_bf16, the strongest trigger).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04c7_lane_probe.txt
: > $O
run() {  # lib-tag victim-op calls aggressor-kind
  timeout 120 python tools/experiments/lane_probe.py aggressor $4 12 > /dev/null 2>&1 &
  PA=$!
  sleep 4
  L=dtlr_amd/libdtlr_hip.so; [ "$1" != "default" ] && L=dtlr_amd/libdtlr_hip_$1.so
  DTLR_HIP_LIB=$PWD/$L timeout 120 python tools/experiments/lane_probe.py victim $2 $3 "lib_$1_vs_$4" 2>&1 | grep -h "victim\|Error" | tee -a $O
  wait $PA
}
for lib in default wait0 g2 occ4; do
  run $lib msda_fused_f32 1500 gemm_bf16
  run $lib msda_fused_f32 1500 mha_f32s
done
for vic in msda_op_f32 msda_fused_bf16 msda_enc_f32 gemm_f32s; do
  run default $vic 400 gemm_bf16
done
