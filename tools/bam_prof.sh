#!/bin/bash
# kernel trace of BAM file -> called variants through the device front (tests/cpp/bam_test --device); run on the GPU box
set -e
out=${1:-gpurun_out/bamprof}
mkdir -p $out
export TMPDIR=/tmp
python tools/bench_bam.py --reads ${READS:-100000} --write-only /tmp/bench.bam ${RICH:+--rich}
timeout -k 10 120 tests/cpp/bam_test /tmp/bench.bam 0 2147483647 --device --time --threads 16 > $out/time.txt
timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bam -- tests/cpp/bam_test /tmp/bench.bam 0 2147483647 --device --time --threads 16 > $out/prof_stdout.txt 2>&1
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
cat $out/time.txt
head -20 $out/kernel_stats.csv
