#!/bin/bash
# cache-side counters of one of the device BAM front's kernels (KERNEL, default bam_decode), separate rocprofv3 --pmc passes; GPU box
set -u
OUT=${1:-gpurun_out/bampmc_mem}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
export TMPDIR=/tmp
[ -f /tmp/bench.bam ] || python3 "$ROOT/tools/bench_bam.py" --reads ${READS:-100000} --write-only /tmp/bench.bam
cd /tmp
i=0
for PASS in \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TA_TCP_STATE_READ_sum" ; do
  i=$((i+1))
  MSQ_BAM_PIECES=1 timeout -s KILL 120 rocprofv3 --pmc $PASS --output-format csv -d "$ROOT/$OUT/pass$i" -o pmc -- "$ROOT/tests/cpp/bam_test" /tmp/bench.bam 0 2147483647 --device --threads 16 > "$ROOT/$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$ROOT/$OUT/pass$i.log"; continue; }
  f=$(find "$ROOT/$OUT/pass$i" -name "*counter_collection.csv" | head -1)
  python3 "$ROOT/tools/pmc_summary.py" "$f" ${KERNEL:-bam_decode}
done
