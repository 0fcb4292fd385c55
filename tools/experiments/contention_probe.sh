#!/bin/bash
# two unlocked copies of contention_probe.py on one GPU; args: forwards-per-copy [env assignments...]
cd ${GRAFT_REPO_ROOT:-.}
N=${1:-200}; shift
env "$@" timeout 600 python tools/experiments/contention_probe.py $N A > gpurun_out/probe_A.log 2>&1 &
PA=$!
env "$@" timeout 600 python tools/experiments/contention_probe.py $N B > gpurun_out/probe_B.log 2>&1 &
PB=$!
wait $PA $PB
grep -h "done\|forward\"\|Error\|error" gpurun_out/probe_A.log gpurun_out/probe_B.log | cut -c1-2200
