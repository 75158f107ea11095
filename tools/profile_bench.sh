#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# Usage: tools/profile_bench.sh <tag> [extra bench args]
set -u
TAG=${1:-r1}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli --no-alt $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $B > $OUT/bench_trace.json 2> $OUT/trace.err
B1="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-alt --no-prof $*"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $B1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $B1 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq -o pmc -- $B1 > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_l2 -o pmc -- $B1 > /dev/null 2> $OUT/pmc_l2.err
for d in trace pmc_fetch pmc_write pmc_sq pmc_l2; do
  f=$(find $OUT/$d -name "*_results.db" | head -1)
  [ -n "$f" ] && python $R/tools/rocpd_summary.py $f > $OUT/$d.txt 2>&1
  tail -3 $OUT/$d.err 2>/dev/null | grep -iE "error|fail" | head -3
done
find $OUT -name "*.db" -size +30M -delete
ls -la $OUT
