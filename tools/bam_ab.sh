#!/bin/bash
# BAM file -> called variants (direct path, 16 threads) with huge / small pages for the inflated stream, interleaved
for r in 1 2 3; do for e in MSQ_BAM_SMALL_PAGES=0 MSQ_BAM_SMALL_PAGES=1; do
  env $e timeout -k 10 300 python tools/bench_bam.py --reads 100000 --threads 16 --call --extra=--direct | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$e', 'inflate', d['inflate_s'], 'index', d['index_s'], 'caller', d['caller_s'], 'first', d['caller_first_s'], 'e2e reads/s', round(d['end_to_end_reads_per_s']))" || exit 1
done; done
