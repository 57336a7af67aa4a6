#!/bin/bash
# rocprofv3 evidence for the multi-scalar multiplication workload: kernel trace + stats, then one PMC pass
# (never combined with tracing).  usage: tools/profile_msm.sh <tag>   (writes under gpurun_out/prof_<tag>_msm/)
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_${TAG}_msm
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --workload msm --steps 5 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES --output-format csv -d $OUT/pmc_SQ -- $BENCH > $OUT/pmc_SQ.log 2>&1 || { echo "pmc failed"; tail -5 $OUT/pmc_SQ.log; exit 1; }
find $OUT -name "*.csv" | head
