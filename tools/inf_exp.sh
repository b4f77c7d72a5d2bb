set -e
export TMPDIR=/tmp
python tools/bench_bam.py --reads 100000 --write-only /tmp/bench.bam
mkdir -p gpurun_out/infexp
for e in 0 1 2 3; do
  MSQ_INF_EXP=$e timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/infexp/t$e -o bam -- tests/cpp/bam_test /tmp/bench.bam 0 2147483647 --device --time --threads 16 > gpurun_out/infexp/out$e.txt 2>&1 || true
  echo "EXP $e"; grep bgzf_inflate gpurun_out/infexp/t$e/bam_kernel_stats.csv | awk -F, '{print $(NF-5)}' 
done
