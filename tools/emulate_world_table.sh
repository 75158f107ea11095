#!/bin/bash
# Rank 0's share of an N-rank job in one process (PFANN_EMULATE_WORLD, no collectives; tuning aid, never a result):
# per-rank step time and scan-kernel time at N = 1, 2, 4, 8 for bench.py's default job (4096 queries per step).
OUT=${1:-gpurun_out/r4/emulate_world.txt}
mkdir -p $(dirname $OUT)
for n in 1 2 4 8; do
  if [ $n = 1 ]; then E="X=1"; else E="PFANN_EMULATE_WORLD=$n"; fi
  env $E python bench.py --steps 4 --warmup 1 --serial --no-cli --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import json,sys
b=json.loads([l for l in sys.stdin if l.startswith('{')][0])
k=b['kernels']
scan={t:v['ms_per_step'] for t,v in k.items() if t.startswith('scan_topk') or t.startswith('topk_')}
gemm=sum(v['ms_per_step'] for t,v in k.items() if t.startswith('conv_gemm'))
tot=sum(scan.values())
print('N=$n rank-step %.2f ms  (GEMM %.2f, scan kernels %.2f, seq_match %.2f, melspec %.2f)  scan_throughput of N such ranks %.3e row pairs/s' % (b['ms_per_step'], gemm, tot, k['seq_match']['ms_per_step'], k['melspec']['ms_per_step'], 1000050.0*77824/(tot*1e-3)))
print('     scan tags (ms per step): ' + ', '.join('%s %.2f' % (t, v) for t, v in sorted(scan.items(), key=lambda kv: -kv[1])))
"
done | tee $OUT
