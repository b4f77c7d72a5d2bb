#!/bin/bash
# experiment builds of the inflate kernel (make LIBDIR=../lib_expN EXTRA=...), each checked against zlib and timed in one
# launch over the 10^5-read file; run on the GPU box
set -e
export TMPDIR=/tmp
python tools/bench_bam.py --reads 100000 --write-only /tmp/bench.bam
mkdir -p gpurun_out/infexp
for v in "" ${VARIANTS:-_exp1 _exp2 _exp3}; do
  L=$PWD/minorseq_amd/lib$v
  [ -f $L/libmsq_hip.so ] || continue
  echo "VARIANT lib$v"
  MSQ_LIB_DIR=$L LD_LIBRARY_PATH=$L timeout -k 10 200 python tools/inf_check.py 3000
  MSQ_BAM_PIECES=1 LD_LIBRARY_PATH=$L timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/infexp/t$v -o bam -- tests/cpp/bam_test /tmp/bench.bam 0 2147483647 --device --time --threads 16 > gpurun_out/infexp/out$v.txt 2>&1
  grep bgzf_inflate gpurun_out/infexp/t$v/bam_kernel_stats.csv | awk -F, '{print "inflate avg ns", $(NF-4)}'
  LD_LIBRARY_PATH=$L timeout -k 10 120 tests/cpp/bam_test /tmp/bench.bam 0 2147483647 --device --time --threads 16
  LD_LIBRARY_PATH=$L timeout -k 10 120 tests/cpp/bam_test /tmp/bench.bam 0 2147483647 --device --time --threads 16
done
