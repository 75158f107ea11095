#!/bin/bash
# A/B/A/B of two builds of the library on one box (tuning aid): gpurun_ab/libA.so, gpurun_ab/libB.so (git-ignored);
# per-layer GEMM times of the encoder at 9728 windows.  tools/ubench/ab_libs.sh [reps]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in $(seq 1 ${1:-2}); do for v in A B; do
  cp gpurun_ab/lib$v.so pfann_amd/libpfann_amd.so
  PFANN_PROF_LAYERS=1 python tools/ubench/embed_rate.py configs/default.json 9728 2>/dev/null | awk -v v=$v '
    /segments\/s/ {printf "%s %s\n", v, $0}
    /conv_gemm_ln_128/ {tot += $(NF-2); if ($0 ~ /K=384/) printf "%s   %s\n", v, $0}
    END {printf "%s   all conv_gemm_ln_128: %.3f ms\n", v, tot}'
done; done
cp gpurun_ab/libA.so pfann_amd/libpfann_amd.so
