#!/bin/bash
# PMC traffic passes on the GPU box (run through gpurun): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 runs (--kernel-trace only),
# reduced to per-shape / per-class bytes per launch by tools/pmc_traffic.py.   usage: bash tools/pmc_pass.sh <tag>
TAG=${1:-vX}
RN=${ROUND:-r06}
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
D=$R/gpurun_out/pmc_$TAG
mkdir -p $D
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d $D/$C -o pmc -- python $R/tools/pmc_traffic.py workload $D/order.json 2>&1 | tail -2 )
  python tools/rocprof_summary.py $D/$C gpurun_out/${RN}_pmc_${C}_step_$TAG --keep > /dev/null 2>&1
done
python tools/pmc_traffic.py reduce $D/FETCH_SIZE $D/WRITE_SIZE $D/order.json gpurun_out/${RN}_step_${TAG}_traffic.json
cp $D/order.json gpurun_out/${RN}_pmc_order_$TAG.json
rm -rf $D
