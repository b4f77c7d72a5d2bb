#!/bin/bash
# the upload pieces of the device BAM front (how many, the first one's share when two), interleaved; GPU box
export TMPDIR=/tmp
python tools/bench_bam.py --reads 100000 --write-only /tmp/bench.bam ${RICH:+--rich}
for i in 1 2 3; do
  for cfg in ${CONFIGS:-"2:40" "2:30" "3:0" "4:0" "5:0"}; do
    n=${cfg%%:*}; p=${cfg##*:}
    echo -n "pieces=$n share=$p "
    MSQ_BAM_PIECES=$n MSQ_BAM_FIRST_SHARE=${p/#0/40} timeout -k 10 120 tests/cpp/bam_test /tmp/bench.bam 0 2147483647 --device --time --threads 16 2>&1 | grep TIMING | tail -1 | sed "s/.*inflate_s/inflate_s/"
  done
done
