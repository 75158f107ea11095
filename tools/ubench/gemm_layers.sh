#!/bin/bash
# per-layer GEMM table for a list of PFANN_T_DBG values (tuning aid)
for v in "$@"; do
  echo "== PFANN_T_DBG=$v"
  PFANN_T_DBG=$v PFANN_PROF_LAYERS=1 python bench.py --no-cpu-baseline --no-alt --steps 3 --filler-db > /tmp/pl.json 2>/dev/null
  python tools/per_layer_table.py /tmp/pl.json | awk '{print $1,$2,$3,$5,$6,$8,$10}' | head -19 | tail -17
done
