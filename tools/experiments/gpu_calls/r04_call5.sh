#!/bin/bash
# Round 4, GPU call 5: victim / aggressor matrix for the "lanes 48..63" effect.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04c5_lane_probe.txt
: > $O
run() {  # victim-op calls aggressor-kind
  timeout 120 python tools/experiments/lane_probe.py aggressor $3 14 > /dev/null 2>&1 &
  PA=$!
  sleep 4
  timeout 120 python tools/experiments/lane_probe.py victim $1 $2 "vs_$3" 2>&1 | grep -h "victim\|Error" | tee -a $O
  wait $PA
}
for agg in none fwd_f32s fwd_f32 gemm_f32s mha_f32s msda_enc_f32 gemm_f32 msda_fused_f32 copy fwd_bf16; do
  run msda_fused_f32 1500 $agg
done
for agg in fwd_f32s msda_enc_f32 mha_f32s; do
  run msda_fused_bf16 1500 $agg
  run msda_enc_f32 300 $agg
  run mha_f32s 600 $agg
done
