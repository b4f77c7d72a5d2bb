#!/bin/bash
# On the GPU box, from the repo root: kernel-trace stats + HBM traffic counters (separate --pmc passes) of bench.py.
# usage: tools/profile_bench.sh <tag> [bench.py args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" "$@" --cpu-sample 0 > "$OUT/stats.log" 2>&1 || { echo "stats pass failed"; tail -5 "$OUT/stats.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" "$@" --steps 3 --warmup 1 --cpu-sample 0 > "$OUT/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/fetch.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" "$@" --steps 3 --warmup 1 --cpu-sample 0 > "$OUT/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/write.log"; exit 1; }
echo ok
