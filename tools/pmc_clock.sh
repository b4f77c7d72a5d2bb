#!/bin/bash
# Effective shader clock and SQ occupancy counters of the pileup kernel for one library build.
# usage (GPU box, repo root): tools/pmc_clock.sh <outdir> <lib dir name> [bench args]
set -u
OUT=$1; LIBD=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
export MSQ_LIB_DIR=$LIBD
cd /tmp && export TMPDIR=/tmp
i=0
for PASS in "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d "$ROOT/$OUT/pass$i" -- python3 "$ROOT/bench.py" "$@" --steps 2 --warmup 1 --cpu-sample 0 > "$ROOT/$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$ROOT/$OUT/pass$i.log"; }
done
echo done
