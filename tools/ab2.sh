#!/bin/bash
# A/B of bench flag sets on one box with one library: tools/ab2.sh <reps> "<common args>" "<flags a>" "<flags b>" ...
REPS=$1; ARGS=$2; shift; shift
for r in $(seq $REPS); do for F in "$@"; do
  timeout -k 10 200 python bench.py --cpu-sample 0 $ARGS $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$F]', round(d['ms_per_step'],4), 'K1', round(r['kernel_ms'],4), 'path', round(r['path_ms'] or 0,4), d.get('verified_against_simple_kernel'))
" || exit 1
done; done
