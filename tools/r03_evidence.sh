#!/bin/bash
# Round-3 evidence on one GPU box, in parts (a gpurun call is at most 20 minutes):
#   tools/r03_evidence.sh lines    the bench lines DESIGN.md quotes, one JSON each
#   tools/r03_evidence.sh dd       the data-dependence sweep (sub-rate rows, minor clones, five strains at 20 % each = the major clone + 4 minors)
#   tools/r03_evidence.sh prof     rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes: default line and 125k shard
#   tools/r03_evidence.sh pmc      SQ counter passes of the default line (pileup_rows_kernel, call_cells_kernel)
# Everything lands in gpurun_out/r03/; tools/r03_collect.sh copies the summaries into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03
mkdir -p "$OUT"
line() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 280 python3 "$ROOT/bench.py" "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "FAILED $name"; tail -3 "$OUT/$name.err"; return 1; }
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
print("%-36s step %.4f ms  K1 %.4f  path %s  phys %s  off-majority %.5f  verified %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r.get("path_ms"), r.get("frac_physical"), d["config"].get("bases_off_majority", 0), d.get("verified_against_simple_kernel")))
PY
}
Q="--cpu-sample 0 --h2d-steps 0 --project-strong 0 --no-compare --verify"
case ${1:-lines} in
lines)
  line bench_default --verify &&
  line bench_pol_125k_force_dist --reads 125000 --force-dist --verify --cpu-sample 0 --steps 200 &&
  line bench_pol_125k_force_dist_unfused --reads 125000 --force-dist --unfused $Q --steps 200 &&
  line bench_default_normalized_layout --layout normalized $Q &&
  line bench_pol_100k --workload pol_100k $Q --steps 200 &&
  line bench_hiv_mix_200k --workload hiv_mix_200k $Q &&
  line bench_amp20k_200k_reads --workload amp20k_10m --reads 200000 $Q &&
  line bench_amp20k_shard_1250k --workload amp20k_10m --reads 1250000 $Q --steps 5 --warmup 2 &&
  line bench_qualqv60_baked --qualqv 60 $Q &&
  line bench_qualqv60_nobake --qualqv 60 --no-bake $Q &&
  line bench_rich_baked --rich $Q &&
  line bench_window_device --window device $Q &&
  line bench_window_host --window host $Q
  ;;
dd)
  line dd_sub0 --sub-rate 0 --n-minors 0 $Q &&
  line dd_sub0_001 --sub-rate 0.001 $Q &&
  line dd_sub0_002 --sub-rate 0.002 $Q &&
  line dd_sub0_005 --sub-rate 0.005 $Q --steps 10 &&
  line dd_sub0_01 --sub-rate 0.01 $Q --steps 5 --warmup 1 &&
  line dd_sub0_02 --sub-rate 0.02 $Q --steps 5 --warmup 1 &&
  line dd_minor10_45codons --n-minors 1 --minor-frac 0.1 --minor-codons 45 $Q &&
  line dd_minor30_45codons --n-minors 1 --minor-frac 0.3 --minor-codons 45 $Q &&
  line dd_minor10_300codons --n-minors 1 --minor-frac 0.1 --minor-codons 300 $Q &&
  line dd_minor30_300codons --n-minors 1 --minor-frac 0.3 --minor-codons 300 $Q &&
  line dd_2minors3pct_300codons --n-minors 2 --minor-frac 0.03 --minor-codons 300 $Q &&
  line dd_4minors1pct_100codons --n-minors 4 --minor-frac 0.01 --minor-codons 100 $Q &&
  line dd_5strains20pct_45codons --n-minors 4 --minor-frac 0.2 --minor-codons 45 $Q --steps 10 &&
  line dd_5strains20pct_100codons --n-minors 4 --minor-frac 0.2 --minor-codons 100 $Q --steps 10 &&
  line dd_5strains20pct_300codons --n-minors 4 --minor-frac 0.2 --minor-codons 300 $Q --steps 5 --warmup 1
  ;;
prof)
  "$ROOT/tools/profile_bench.sh" r03/prof_default --no-compare --h2d-steps 0 --project-strong 0 &&
  "$ROOT/tools/profile_bench.sh" r03/prof_125k --reads 125000 --force-dist --steps 100 --no-compare --h2d-steps 0 --project-strong 0
  ;;
pmc)
  "$ROOT/tools/pmc_run.sh" gpurun_out/r03/pmc_default --no-compare --h2d-steps 0 --project-strong 0
  ;;
esac
