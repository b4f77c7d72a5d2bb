#!/bin/bash
# A/B of environment settings on one box: tools/ab_env.sh <reps> "<bench args>" "VAR=a" "VAR=b" ...
REPS=$1; ARGS=$2; shift; shift
for r in $(seq $REPS); do for E in "$@"; do
  env $E timeout -k 10 200 python bench.py --cpu-sample 0 $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$E]', round(d['ms_per_step'],4), 'K1', round(r['kernel_ms'],4), 'path', round(r['path_ms'] or 0,4))
" || exit 1
done; done
