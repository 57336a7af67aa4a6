#!/bin/bash
# On the GPU box: parity check (sizes that take the large-radix kernels) of each tools/bin/NAME.so given, then the A/B timing.
mkdir -p gpurun_out
for name in "$@"; do
  PLONK_HIP_LIB=$PWD/tools/bin/$name.so timeout -k 10 300 python -u tools/quick_check.py 131072 262144 1048576 > gpurun_out/abcheck_$name.txt 2>&1 || { echo "$name: check failed"; tail -n 5 gpurun_out/abcheck_$name.txt; exit 1; }
  tail -n 1 gpurun_out/abcheck_$name.txt
done
bash tools/ab_run.sh "$@"
