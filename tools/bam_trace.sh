#!/bin/bash
# phase times of the device BAM front (MSQ_BAM_TRACE) on the synthetic 10^5-read file; run on the GPU box
set -e
export TMPDIR=/tmp
python tools/bench_bam.py --reads ${READS:-100000} --write-only /tmp/bench.bam ${RICH:+--rich}
for i in 1 2; do MSQ_BAM_TRACE=1 timeout -k 10 120 tests/cpp/bam_test /tmp/bench.bam 0 2147483647 --device --time --threads 16; done
