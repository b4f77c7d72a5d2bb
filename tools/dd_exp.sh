#!/bin/bash
# where K1's time goes on noisy data: the shipped kernel, a timing build without the drain's global atomics
# (MSQ_ROWS_EXP=1: counts are wrong) and one without the deviation queue at all (=2); GPU box
for rate in 0.005 0.01; do
  for lib in lib lib_exp1 lib_exp2; do
    [ -f minorseq_amd/$lib/libmsq_hip.so ] || continue
    MSQ_LIB_DIR=$lib timeout -k 10 200 python bench.py --cpu-sample 0 --sub-rate $rate --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('rate $rate %-9s step %.4f K1 %.4f' % ('$lib', d['ms_per_step'], r['kernel_ms']))
"
  done
done
