#!/bin/bash
# Round 4, GPU call 8: call 7 again with the box identified (the effect was absent on call 7's box, present on the four before it).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04c8_lane_probe.txt
{ hostname; uname -r; cat /sys/module/amdgpu/version 2>/dev/null; rocm-smi --showfwinfo 2>/dev/null | grep -iE "MEC|SMC|SDMA|RLC|CP|VBIOS" | head -12; rocm-smi --showuniqueid --showserial 2>/dev/null | grep -iE "unique|serial" | head -4; } > $O 2>&1
python -c "import torch; print(torch.cuda.get_device_properties(0))" >> $O 2>&1
run() {  # lib-tag victim-op calls aggressor-kind
  timeout 120 python tools/experiments/lane_probe.py aggressor $4 14 > /dev/null 2>&1 &
  PA=$!
  sleep 5
  L=dtlr_amd/libdtlr_hip.so; [ "$1" != "default" ] && L=dtlr_amd/libdtlr_hip_$1.so
  DTLR_HIP_LIB=$PWD/$L timeout 120 python tools/experiments/lane_probe.py victim $2 $3 "lib_$1_vs_$4" 2>&1 | grep -h "victim\|Error" | tee -a $O
  wait $PA
}
run default msda_fused_f32 200 none          # warms the page cache (the first torch import of a box takes a minute)
for lib in default wait0 g2 occ4 default; do
  run $lib msda_fused_f32 1500 gemm_bf16
done
run default msda_fused_f32 1500 mha_f32s
run g2 msda_fused_f32 1500 mha_f32s
run wait0 msda_fused_f32 1500 mha_f32s
head -24 $O
