#!/bin/bash
# the first piece's share of the file (two-piece upload of the device BAM front), interleaved; GPU box
export TMPDIR=/tmp
python tools/bench_bam.py --reads 100000 --write-only /tmp/bench.bam
for i in 1 2 3; do for p in 33 45 55 65; do echo -n "share=$p "; MSQ_BAM_FIRST_SHARE=$p MSQ_BAM_TRACE=1 timeout -k 10 120 tests/cpp/bam_test /tmp/bench.bam 0 2147483647 --device --time --threads 16 2>&1 | grep TIMING | tail -1 | sed "s/.*inflate_s/inflate_s/"; done; done
