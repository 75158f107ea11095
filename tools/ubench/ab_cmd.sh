#!/bin/bash
# A/B/A/B of two builds of the library (gpurun_ab/libA.so, libB.so) under any command: tools/ubench/ab_cmd.sh REPS cmd...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
reps=$1; shift
for i in $(seq 1 $reps); do for v in A B; do
  cp gpurun_ab/lib$v.so pfann_amd/libpfann_amd.so
  "$@" 2>/dev/null | sed "s/^/$v /"
done; done
cp gpurun_ab/libA.so pfann_amd/libpfann_amd.so
