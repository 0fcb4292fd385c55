#!/bin/bash
# repeat the 2-rank single-GPU data-parallel self-check of bench.py under different kernel switches (which kernel breaks bit-reproducibility?)
cd ${GRAFT_REPO_ROOT:-.}
run() {
  P=$((20000 + RANDOM % 20000))
  env "$@" timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 2 --steps 2 --warmup 1 --batch 3 --backend gloo --single-device --no-cpu-baseline --no-parity --min-seconds 0 2>&1 | grep -o "dp_verified = [A-Za-z]*" | sed 's/dp_verified = //' | tr '\n' ' '
}
for cfg in "$@"; do
  echo -n "$cfg : "
  for i in 1 2 3 4 5 6; do run $cfg; done
  echo
done
