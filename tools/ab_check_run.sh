#!/bin/bash
# On the GPU box: parity check of each tools/bin/NAME.so given (sizes that take one, two and three passes, small and
# large radices), then the A/B timing of the headline step.
mkdir -p gpurun_out
for name in "$@"; do
  PLONK_HIP_LIB=$PWD/tools/bin/$name.so timeout -k 10 400 python -u tools/quick_check.py 4 64 2048 4096 8192 65536 131072 262144 1048576 > gpurun_out/abcheck_$name.txt 2>&1 || { echo "$name: check failed"; tail -n 12 gpurun_out/abcheck_$name.txt; exit 1; }
  tail -n 1 gpurun_out/abcheck_$name.txt
done
bash tools/ab_run.sh "$@"
