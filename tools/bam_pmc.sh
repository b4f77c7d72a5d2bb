#!/bin/bash
# SQ counters of the device BAM front's kernels (separate rocprofv3 --pmc passes over tests/cpp/bam_test --device); GPU box
set -u
OUT=${1:-gpurun_out/bampmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
export TMPDIR=/tmp
[ -f /tmp/bench.bam ] || python3 "$ROOT/tools/bench_bam.py" --reads ${READS:-100000} --write-only /tmp/bench.bam
cd /tmp
i=0
for PASS in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" ; do
  i=$((i+1))
  MSQ_BAM_PIECES=1 timeout -s KILL 120 rocprofv3 --pmc $PASS --output-format csv -d "$ROOT/$OUT/pass$i" -o pmc -- "$ROOT/tests/cpp/bam_test" /tmp/bench.bam 0 2147483647 --device --threads 16 > "$ROOT/$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$ROOT/$OUT/pass$i.log"; exit 1; }
  f=$(find "$ROOT/$OUT/pass$i" -name "*counter_collection.csv" | head -1)
  python3 "$ROOT/tools/pmc_summary.py" "$f" ${KERNEL:-bgzf_inflate}
done
