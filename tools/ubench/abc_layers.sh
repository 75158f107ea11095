#!/bin/bash
# tools/ubench/abc_layers.sh "A L C" REPS : per-layer conv GEMM times of the encoder at 9728 windows under each library build
cd ${GRAFT_REPO_ROOT:-/root/repo}
vs=$1; reps=$2
for i in $(seq 1 $reps); do for v in $vs; do
  cp gpurun_ab/lib$v.so pfann_amd/libpfann_amd.so
  PFANN_PROF_LAYERS=1 python tools/ubench/embed_rate.py configs/default.json 9728 2>/dev/null | awk -v v=$v '
    /conv_gemm_ln_128/ {tot += $(NF-2); if ($0 ~ /K=384/) k384 += $(NF-2)}
    END {printf "%s   all conv_gemm_ln_128: %.3f ms   K=384 layers: %.3f ms\n", v, tot, k384}'
done; done
cp gpurun_ab/libA.so pfann_amd/libpfann_amd.so
