#!/bin/bash
# round-3 evidence of the BAM fronts (host direct decode vs device front), run on the GPU box from the repo root
set -e
out=gpurun_out/r03bam
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python tools/bench_bam.py --reads 100000 --threads 16 --call --extra=--direct --out $out/r03_bam_front_100k_direct.json
timeout -k 10 300 python tools/bench_bam.py --reads 100000 --threads 16 --call --extra=--device --out $out/r03_bam_front_100k_device.json
timeout -k 10 300 python tools/bench_bam.py --reads 50000 --rich --threads 16 --call --extra=--direct --out $out/r03_bam_front_50k_rich_direct.json
timeout -k 10 300 python tools/bench_bam.py --reads 50000 --rich --threads 16 --call --extra=--device --out $out/r03_bam_front_50k_rich_device.json
bash tools/bam_prof.sh $out/prof > $out/prof.txt 2>&1
cp $out/prof/kernel_stats.csv $out/r03_bam_front_100k_device_kernel_stats.csv
MSQ_BAM_TRACE=1 tests/cpp/bam_test /tmp/bench.bam 0 2147483647 --device --time --threads 16 > $out/r03_bam_front_100k_device_trace.txt 2>&1
MSQ_BAM_PIECES=1 MSQ_BAM_TRACE=1 tests/cpp/bam_test /tmp/bench.bam 0 2147483647 --device --time --threads 16 > $out/r03_bam_front_100k_device_trace_one_piece.txt 2>&1
