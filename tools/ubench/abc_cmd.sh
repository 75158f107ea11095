#!/bin/bash
# A/B/C... of builds gpurun_ab/lib{A,B,C}.so under any command: tools/ubench/abc_cmd.sh "A B C" REPS cmd...
cd ${GRAFT_REPO_ROOT:-/root/repo}
vs=$1; reps=$2; shift; shift
for i in $(seq 1 $reps); do for v in $vs; do
  cp gpurun_ab/lib$v.so pfann_amd/libpfann_amd.so
  "$@" 2>/dev/null | sed "s/^/$v /"
done; done
cp gpurun_ab/libA.so pfann_amd/libpfann_amd.so
