#!/bin/bash
# per-layer table of the fused conv GEMM (tuning aid): tools/ubench/gemm_layers.sh [extra bench args]
PFANN_PROF_LAYERS=1 python bench.py --no-cpu-baseline --no-cli --no-alt --steps 3 --filler-db "$@" > /tmp/pl.json 2>/dev/null
python tools/per_layer_table.py /tmp/pl.json
