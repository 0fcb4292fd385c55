#!/bin/bash
# Round 4, GPU call 6: which aggressors trigger, which victims are vulnerable.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04c6_lane_probe.txt
: > $O
run() {  # victim-op calls aggressor-kind
  timeout 120 python tools/experiments/lane_probe.py aggressor $3 12 > /dev/null 2>&1 &
  PA=$!
  sleep 4
  timeout 120 python tools/experiments/lane_probe.py victim $1 $2 "vs_$3" 2>&1 | grep -h "victim\|Error" | tee -a $O
  wait $PA
}
for agg in mha_bf16 gemm_bf16 gemm_bf16_f32out gemm_f16 ffn_bf16 msda_enc_bf16 topk mha_f32; do
  run msda_fused_f32 1500 $agg
done
for vic in msda_op_f32 gemm_f32s ln_f32 gemm_bf16 ffn_bf16 mha_bf16 msda_enc_bf16 gemm_f32 mha_f32; do
  run $vic 400 mha_f32s
done
