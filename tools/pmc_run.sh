#!/bin/bash
# Collect SQ counters for the pileup kernel in separate rocprofv3 --pmc passes (one run per pass).
# usage (on the GPU box, from the repo root): tools/pmc_run.sh <outdir> <bench.py args...>
set -u
OUT=$1; shift
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for PASS in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
  "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_VALU" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $PASS --output-format csv -d "$ROOT/$OUT/pass$i" -- python3 "$ROOT/bench.py" "$@" --steps 2 --warmup 1 --cpu-sample 0 > "$ROOT/$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$ROOT/$OUT/pass$i.log"; exit 1; }
done
echo done
