#!/bin/bash
# The wide-window lines of tools/r02_evidence.sh again (merged slab launches), with the launches per slab beside them.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02
mkdir -p "$OUT"
line() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 280 python3 "$ROOT/bench.py" "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "FAILED $name"; tail -3 "$OUT/$name.err"; return 1; }
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
print("%-40s step %.4f ms  K1 %.4f  path %s  frac %.3f  launches/step %s  verified %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r.get("path_ms"), r["frac"], d["config"].get("pileup_launches_per_step"), d.get("verified_against_simple_kernel")))
PY
}
line bench_hiv_mix_200k --workload hiv_mix_200k --verify --cpu-sample 0 &&
MSQ_NO_MERGE=1 line bench_hiv_mix_200k_launch_per_slab --workload hiv_mix_200k --verify --cpu-sample 0 &&
line bench_amp20k_200k_reads --workload amp20k_10m --reads 200000 --verify --cpu-sample 0 &&
MSQ_NO_MERGE=1 line bench_amp20k_200k_reads_launch_per_slab --workload amp20k_10m --reads 200000 --verify --cpu-sample 0 &&
line bench_hiv_mix_200k_no_partition --workload hiv_mix_200k --no-partition --verify --cpu-sample 0 &&
line bench_default_check --verify --cpu-sample 0
