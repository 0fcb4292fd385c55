#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
N=${1:-60}; MODES=${2:-none,all,msda_args,msda_out,lin_out,not_msda,none}; shift; shift
env "$@" timeout 800 python tools/experiments/reuse_probe.py $N A $MODES > gpurun_out/reuse_A.log 2>&1 &
PA=$!
env "$@" timeout 800 python tools/experiments/reuse_probe.py $N B $MODES > gpurun_out/reuse_B.log 2>&1 &
PB=$!
wait $PA $PB
grep -h "mode" gpurun_out/reuse_A.log gpurun_out/reuse_B.log
