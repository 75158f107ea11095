for cfg in "1 1" "1 2" "1 3" "1 4" "0 1" "0 2" "0 3"; do
  set -- $cfg
  if [ "$1" = "1" ]; then export PFANN_NO_FUSE=1; else unset PFANN_NO_FUSE; fi
  export PFANN_STREAMS=$2
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('nofuse=$1 streams=$2', o['value'], o['ms_per_step'], o['top1_hit_rate'])"
done
