#!/bin/bash
# decode kernel variants (make LIBDIR=../lib_expN EXTRA=-DMSQ_DECODE_UNROLL=..): kernel time on the 10^5-read file; GPU box
export TMPDIR=/tmp
python tools/bench_bam.py --reads 100000 --write-only /tmp/bench.bam
for v in "" ${VARIANTS:-_exp1 _exp2}; do
  L=$PWD/minorseq_amd/lib$v
  [ -f $L/libmsq_hip.so ] || continue
  LD_LIBRARY_PATH=$L timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/decexp/t$v -o bam -- tests/cpp/bam_test /tmp/bench.bam 0 2147483647 --device --time --threads 16 > gpurun_out/decexp_out$v.txt 2>&1
  echo "lib$v: $(grep bam_decode gpurun_out/decexp/t$v/bam_kernel_stats.csv | awk -F, '{print $(NF-4)}') ns"
done
