for d in 0 1 2 3; do
  export PFANN_DBG=$d
  python bench.py --queries 256 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); k=o['kernels']['conv_gemm_ln_128']; print('dbg=$d', o['ms_per_step'], k['ms_per_step'], k.get('TFLOPs'))"
done
