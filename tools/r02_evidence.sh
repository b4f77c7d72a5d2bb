#!/bin/bash
# Round-2 evidence on one GPU box: the bench lines DESIGN.md quotes (one JSON each) + rocprofv3 kernel stats and HBM
# traffic counters of the default line.  usage (on the box, repo root): tools/r02_evidence.sh
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
print("%-34s step %.4f ms  K1 %.4f  path %s  frac %.3f  phys %s  verified %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r.get("path_ms"), r["frac"], r.get("frac_physical"), d.get("verified_against_simple_kernel")))
PY
}
line bench_default --verify &&
line bench_default_h2d --h2d-steps 5 --cpu-sample 0 --verify &&
line bench_pol_125k_force_dist --reads 125000 --force-dist --verify --cpu-sample 0 --steps 200 &&
line bench_pol_125k_force_dist_unfused --reads 125000 --force-dist --unfused --verify --cpu-sample 0 --steps 200 &&
line bench_pol_100k --workload pol_100k --verify --cpu-sample 0 --steps 200 &&
line bench_hiv_mix_200k --workload hiv_mix_200k --verify --cpu-sample 0 &&
line bench_amp20k_200k_reads --workload amp20k_10m --reads 200000 --verify --cpu-sample 0 &&
line bench_qualqv60_baked --qualqv 60 --verify --cpu-sample 0 &&
line bench_qualqv60_nobake --qualqv 60 --no-bake --verify --cpu-sample 0 &&
line bench_rich_baked --rich --verify --cpu-sample 0 &&
line bench_minor10_45codons --n-minors 1 --minor-frac 0.1 --minor-codons 45 --verify --cpu-sample 0 &&
line bench_minor30_45codons --n-minors 1 --minor-frac 0.3 --minor-codons 45 --verify --cpu-sample 0 &&
line bench_minor10_300codons --n-minors 1 --minor-frac 0.1 --minor-codons 300 --verify --cpu-sample 0 &&
line bench_minor30_300codons --n-minors 1 --minor-frac 0.3 --minor-codons 300 --verify --cpu-sample 0 &&
line bench_2minors3pct_300codons --n-minors 2 --minor-frac 0.03 --minor-codons 300 --verify --cpu-sample 0 &&
line bench_sub2pct --sub-rate 0.02 --verify --cpu-sample 0 --steps 5 --warmup 1 &&
line bench_window_device --window device --verify --cpu-sample 0 &&
line bench_window_host --window host --verify --cpu-sample 0 &&
"$ROOT/tools/profile_bench.sh" r02/prof_default &&
"$ROOT/tools/profile_bench.sh" r02/prof_125k --reads 125000 --force-dist --steps 100
