#!/bin/bash
# Kernel timing experiments: build libmsq_hip.so with -DMSQ_EXP=<n> (see msq_pileup_stream.hip; n != 0 gives WRONG
# counts, timing only), run bench.py on a GPU box, and ALWAYS finish with a clean rebuild of the shipped library.
# A variant that fails to build is skipped: the stale library is never timed.
# usage (in the build container, repo root): tools/exp_run.sh <n> [<n> ...] [-- bench.py args]
set -u
cd "$(dirname "$0")/.."
SRC=minorseq_amd/csrc
restore() { touch $SRC/msq_pileup_stream.hip; make -s -C $SRC EXTRA= >/dev/null 2>&1 && echo "clean library rebuilt" || echo "CLEAN REBUILD FAILED"; }
trap restore EXIT
VARS=(); ARGS=()
while [ $# -gt 0 ]; do [ "$1" = "--" ] && { shift; ARGS=("$@"); break; }; VARS+=("$1"); shift; done
mkdir -p gpurun_out/exp
for E in "${VARS[@]}"; do
  touch $SRC/msq_pileup_stream.hip
  if ! make -s -C $SRC EXTRA=-DMSQ_EXP=$E > gpurun_out/exp/build$E.log 2>&1; then echo "EXP $E: build failed, skipped (gpurun_out/exp/build$E.log)"; continue; fi
  /usr/local/graft/bin/gpurun --timeout 300 -- "mkdir -p gpurun_out/exp && timeout -k 10 150 python bench.py --cpu-sample 0 ${ARGS[*]:-} > gpurun_out/exp/e$E.json 2> gpurun_out/exp/e$E.err && python3 -c \"import json;d=json.load(open('gpurun_out/exp/e$E.json'));print('EXP $E kernel_ms %.4f ms_per_step %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']))\"" 2>&1 | grep "^EXP\|status=fail\|rc=[1-9]"
done
