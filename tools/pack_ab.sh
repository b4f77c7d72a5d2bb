#!/bin/bash
# AminoAcidCaller's constructor through the C++ front (tests/cpp/bam_test --call): compact vs record staging, interleaved.
# usage: tools/pack_ab.sh [bench_bam.py arguments, e.g. --extra=--direct --rich --reads 50000]
for r in 1 2 3; do for e in MSQ_PACK_WIDE_STAGING=0 MSQ_PACK_WIDE_STAGING=1 MSQ_X=unset; do
  env $e timeout -k 10 300 python tools/bench_bam.py --threads 16 --call "$@" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$e $*', 'caller', d['caller_s'], 'first', d['caller_first_s'], 'e2e', round(d['end_to_end_reads_per_s']))" || exit 1
done; done
