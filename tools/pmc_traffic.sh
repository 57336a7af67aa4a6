#!/bin/bash
# On the GPU box: memory-side traffic per kernel (FETCH_SIZE, WRITE_SIZE: one rocprofv3 --pmc pass each, no tracing) of
# bench.py's step for each tools/bin/NAME.so given.  Output: gpurun_out/traffic_NAME.txt (KiB per launch, averaged).
set -u
export TMPDIR=/tmp
for name in "$@"; do
  export PLONK_HIP_LIB=$PWD/tools/bin/$name.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_${name}_$c
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${name}_$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-lean-variant --no-profile > gpurun_out/pmc_${name}_$c.log 2>&1 || { echo "$name $c failed"; tail -3 gpurun_out/pmc_${name}_$c.log; exit 1; }
  done
  python3 - "$name" <<'PY' > gpurun_out/traffic_$name.txt
import csv, glob, re, sys
from collections import defaultdict
name = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_{name}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"plk(_field\d+)?::", "", r["Kernel_Name"].split("(")[0]).replace("void ", "")
            if "table_kernel" in k or "ntt" not in k and "gp_" not in k and "constraint" not in k:
                continue
            acc[(k, r["Grid_Size"])][c].append(float(r["Counter_Value"]))
print(f"{name}: KiB per launch (FETCH_SIZE raw, WRITE_SIZE), traffic = (2 FETCH + WRITE) KiB -> GB")
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]["FETCH_SIZE"])):
    f = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1); w = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
    print(f"{k[:60]:60s} grid {g:>9s} x{len(v['FETCH_SIZE']):2d}  fetch {f:12.0f}  write {w:12.0f}  traffic {(2 * f + w) * 1024 / 1e9:7.3f} GB")
PY
  cat gpurun_out/traffic_$name.txt
done
