#!/bin/bash
# On the GPU box: time each tools/bin/NAME.so given, twice, alternating.  Output: gpurun_out/ab.txt
mkdir -p gpurun_out; : > gpurun_out/ab.txt
for rep in 1 2; do
  for name in "$@"; do
    PLONK_HIP_LIB=$PWD/tools/bin/$name.so timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || { echo "$name failed" >> gpurun_out/ab.txt; exit 1; }
    python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab_$name.json')); print('$name', d['ms_per_step'], ' '.join('%s=%.3f'%(k['label'].replace('_kernel',''),k['ms_per_step']) for k in d['kernels']))" >> gpurun_out/ab.txt
  done
done
cat gpurun_out/ab.txt
