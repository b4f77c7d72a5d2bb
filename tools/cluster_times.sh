#!/bin/bash
# gather + clustering times of tools/bench_phase.py for several numbers of positions (pol_1m)
for k in "$@"; do
  timeout -k 10 200 python tools/bench_phase.py --steps 20 --positions $k | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['positions'], 'gather', round(d['ms'],4), 'cluster', round(d['cluster_ms'],4), d['clusters'])" || exit 1
done
