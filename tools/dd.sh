#!/bin/bash
# data-dependence sweep of the pileup kernel on one box (each line verified against the simple kernel)
# usage: tools/dd.sh [lib dir]
export MSQ_LIB_DIR=${1:-lib}
run() { timeout -k 10 200 python bench.py --cpu-sample 0 --verify "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-62s step %.4f K1 %.4f off-majority %.4f %s' % ('$*', d['ms_per_step'], r['kernel_ms'], d['config']['bases_off_majority'], d.get('verified_against_simple_kernel')))
" || exit 1; }
run &&
run --n-minors 1 --minor-frac 0.1 --minor-codons 45 &&
run --n-minors 1 --minor-frac 0.3 --minor-codons 45 &&
run --n-minors 1 --minor-frac 0.1 --minor-codons 300 &&
run --n-minors 1 --minor-frac 0.3 --minor-codons 300 &&
run --n-minors 2 --minor-frac 0.03 --minor-codons 300 &&
run --n-minors 4 --minor-frac 0.2 --minor-codons 100 &&
run --n-minors 4 --minor-frac 0.01 --minor-codons 100 &&
run --sub-rate 0.005 --steps 5 &&
run --sub-rate 0.02 --steps 5 --warmup 1 &&
run --reads 125000 --force-dist --steps 200 &&
run --workload hiv_mix_200k &&
run --qualqv 60
