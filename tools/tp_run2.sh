#!/bin/bash
# On the GPU box: rounds 2+3 at several sizes and the plain transform at 2^22 under the two-pass knob.
for l in 17 18 19; do for k in 16 20 16 20; do
  PLONK_NTT_TWO_PASS_MAX_LOG=$k timeout -k 10 200 python -u bench.py --log-n $l --steps 20 --warmup 3 --no-cpu-baseline --no-lean-variant --no-profile > gpurun_out/tp2.json 2> gpurun_out/tp2.err || { echo "failed"; tail -n 3 gpurun_out/tp2.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/tp2.json')); print('round23 n=2^$l two-pass<=2^$k', d['ms_per_step'])"
done; done
for l in 21 22; do for k in 16 22 16 22; do
  PLONK_NTT_TWO_PASS_MAX_LOG=$k timeout -k 10 200 python -u bench.py --workload ntt --log-n $l --steps 20 --warmup 3 --no-cpu-baseline --no-profile > gpurun_out/tp2.json 2> gpurun_out/tp2.err || { echo "failed"; tail -n 3 gpurun_out/tp2.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/tp2.json')); print('ntt n=2^$l two-pass<=2^$k', d['ms_per_step'], d.get('config',{}).get('direct_transform_ms'))"
done; done
