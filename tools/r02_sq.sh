#!/bin/bash
# SQ counters of the default bench line (three passes, tools/pmc_run.sh) and the LDS counters of the kernel variants
# with one phase removed (MSQ_EXP builds under minorseq_amd/lib_exp<n>: timing / counters only, wrong counts).
# usage (on the GPU box, repo root): tools/r02_sq.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
"$ROOT/tools/pmc_run.sh" gpurun_out/r02_sq/default || exit 1
cd /tmp && export TMPDIR=/tmp
for E in 1 2 16; do
  [ -f "$ROOT/minorseq_amd/lib_exp$E/libmsq_hip.so" ] || continue
  export MSQ_LIB_DIR=lib_exp$E
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$ROOT/gpurun_out/r02_sq/exp$E" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --cpu-sample 0 > "$ROOT/gpurun_out/r02_sq/exp$E.log" 2>&1 || { echo "exp $E failed"; tail -3 "$ROOT/gpurun_out/r02_sq/exp$E.log"; exit 1; }
done
unset MSQ_LIB_DIR
echo done
