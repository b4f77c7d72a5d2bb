#!/bin/bash
# A/B on one box: bench.py with each library build named (dirs under minorseq_amd/), interleaved, `reps` times.
# usage: tools/ab.sh <reps> "<bench args>" lib_a lib_b ...
REPS=$1; ARGS=$2; shift; shift
for r in $(seq $REPS); do for L in "$@"; do
  MSQ_LIB_DIR=$L timeout -k 10 200 python bench.py --cpu-sample 0 $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$L', round(d['ms_per_step'],4), 'K1', round(r['kernel_ms'],4), 'path', round(r['path_ms'] or 0,4), d.get('verified_against_simple_kernel'))
" || exit 1
done; done
