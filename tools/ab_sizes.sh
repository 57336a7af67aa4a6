#!/bin/bash
# On the GPU box: sizes other than the headline under each tools/bin/NAME.so given - direct transforms of 2^22 and 2^24 points,
# the quotient step, rounds 2+3 at n = 2^16 and 2^18 - twice, alternating.  Output: gpurun_out/ab_sizes.txt
mkdir -p gpurun_out; : > gpurun_out/ab_sizes.txt
run() {  # name, label, bench arguments...
  local name=$1 label=$2; shift 2
  PLONK_HIP_LIB=$PWD/tools/bin/$name.so timeout -k 10 200 python -u bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-profile > gpurun_out/abs.json 2> gpurun_out/abs.err || { echo "$name $label failed" >> gpurun_out/ab_sizes.txt; tail -n 3 gpurun_out/abs.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/abs.json')); print('$name', '$label', 'ms_per_step', d.get('ms_per_step'), 'direct', d.get('direct_transform_ms'))" >> gpurun_out/ab_sizes.txt
}
for rep in 1 2; do
  for name in "$@"; do
    run $name ntt22 --workload ntt --log-n 22
    run $name ntt24 --workload ntt --log-n 24
    run $name quotient --workload quotient
    run $name r23_16 --log-n 16
    run $name r23_18 --log-n 18
  done
done
cat gpurun_out/ab_sizes.txt
