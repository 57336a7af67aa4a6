#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box:
#   kernel trace + stats, then one run per PMC group (never combined with tracing).
# usage: tools/profile.sh <tag>      (writes under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-lean-variant"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -- $BENCH > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $OUT/pmc_$name.log; exit 1; }
done
find $OUT -name "*.csv" | head -30
