#!/bin/bash
# On the GPU box: the direct transform at 2^21 .. 2^22 and the quotient step under the two-pass knob.
for l in 21 22; do for k in 16 22 16 22; do
  PLONK_NTT_TWO_PASS_MAX_LOG=$k timeout -k 10 200 python -u bench.py --workload ntt --log-n $l --steps 20 --warmup 3 --no-cpu-baseline --no-profile > gpurun_out/tp2.json 2> gpurun_out/tp2.err || { echo "failed"; tail -n 3 gpurun_out/tp2.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/tp2.json')); print('ntt n=2^$l two-pass<=2^$k direct', d.get('direct_transform_ms'))"
done; done
for k in 16 22 16 22; do
  PLONK_NTT_TWO_PASS_MAX_LOG=$k timeout -k 10 200 python -u bench.py --workload quotient --steps 20 --warmup 3 --no-cpu-baseline --no-profile > gpurun_out/tp2.json 2> gpurun_out/tp2.err || { echo "failed"; tail -n 3 gpurun_out/tp2.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/tp2.json')); print('quotient two-pass<=2^$k', d.get('ms_per_step'))"
done
