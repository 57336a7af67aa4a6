#!/bin/bash
# On the GPU box: time bench.py under the transform-plan knobs given as "TWO_PASS_MAX_LOG:FIRST_RADIX_LOG" words.
for spec in "$@"; do
  k=${spec%%:*}; f=${spec#*:}
  PLONK_NTT_TWO_PASS_MAX_LOG=$k PLONK_NTT_FIRST_RADIX_LOG=$f timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-lean-variant > gpurun_out/tp_$k_$f.json 2> gpurun_out/tp.err || { echo "$spec failed"; tail -n 3 gpurun_out/tp.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/tp_$k_$f.json')); print('$spec', d['ms_per_step'], ' '.join('%s=%.3f'%(k['label'].replace('_kernel',''),k['ms_per_step']) for k in d['kernels'] if 'side' not in k['label'] and 'gp_' not in k['label']))"
done
