#!/bin/bash
# On the GPU box: memory-side bytes read and written by the two launches of the extension (tools/ntt_harness.hip built
# as tools/bin/ntt_harness_cur) against the number of columns: one rocprofv3 --pmc pass per counter, no tracing.
#   COLS="1 8 17" PAD=128 bash tools/fetch_by_columns.sh      (PAD: bytes of skew per vector, default 0)
set -u
export TMPDIR=/tmp
B=$PWD/tools/bin/${HARNESS:-ntt_harness_cur}
export COLS="${COLS:-17}"
OUT=$PWD/gpurun_out
cd /tmp
for cols in $COLS; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/fc_${cols}_$c
    timeout -s KILL 100 rocprofv3 --pmc $c --output-format csv -d $OUT/fc_${cols}_$c -- $B $cols ${PAD:-0} > $OUT/fc_${cols}_$c.log 2>&1 || { echo "cols $cols $c failed"; tail -3 $OUT/fc_${cols}_$c.log; exit 1; }
  done
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for cols in [int(x) for x in os.environ["COLS"].split()]:
    acc = defaultdict(lambda: defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{out}/fc_{cols}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = "pass" if "ntt_pass" in r["Kernel_Name"] else "final" if "ntt_final" in r["Kernel_Name"] else None
                if k: acc[k][c].append(float(r["Counter_Value"]))
    for k in ("pass", "final"):
        f = sum(acc[k]["FETCH_SIZE"]) / max(1, len(acc[k]["FETCH_SIZE"])); w = sum(acc[k]["WRITE_SIZE"]) / max(1, len(acc[k]["WRITE_SIZE"]))
        print(f"cols {cols:2d} {k:5s}: fetch 2x{f:10.0f} KiB = {2 * f * 1024 / 1e6:8.1f} MB ({2 * f * 1024 / 1e6 / cols:6.1f} per column)   write {w * 1024 / 1e6:8.1f} MB ({w * 1024 / 1e6 / cols:6.1f} per column)")
PY
