#!/bin/bash
# Copy what tools/r03_evidence.sh left under gpurun_out/r03 into profiles/ (tracked): bench lines, kernel stats, PMC summaries.
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/r03
for f in $S/bench_*.json $S/dd_*.json; do [ -s "$f" ] && cp "$f" profiles/r03_$(basename "$f"); done
ks() { find "$S/$1/stats" -name "*kernel_stats.csv" | head -1; }
cc() { find "$S/$1" -name "*counter_collection.csv" | head -1; }
[ -d $S/prof_default ] && cp "$(ks prof_default)" profiles/r03_bench_pol1m_default_kernel_stats.csv
[ -d $S/prof_125k ] && cp "$(ks prof_125k)" profiles/r03_bench_pol125k_force_dist_kernel_stats.csv
if [ -d $S/prof_default ]; then
  {
    echo "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/r03_evidence.sh prof; KB per dispatch; gfx950: FETCH_SIZE counts half the bytes of a 16 B/lane stream)"
    echo "== bench.py (pol_1m, 1M reads, column-aligned set), --steps 3 --warmup 1"
    python3 tools/pmc_summary.py "$(cc prof_default/fetch)"; python3 tools/pmc_summary.py "$(cc prof_default/write)"
    echo "== bench.py --reads 125000 --force-dist (the strong-scaling shard)"
    python3 tools/pmc_summary.py "$(cc prof_125k/fetch)"; python3 tools/pmc_summary.py "$(cc prof_125k/write)"
  } | python3 -c "
import sys
keep = False
for l in sys.stdin:
    if not l.startswith('   '):
        keep = l.startswith('==') or l.startswith('rocprofv3') or any(k in l for k in ('pileup_rows', 'fold_partials', 'count_tests', 'call_cells'))
    if keep:
        sys.stdout.write(l)
" > profiles/r03_pmc_hbm_traffic.txt
fi
if [ -d $S/pmc_default ]; then
  for k in pileup_rows call_cells; do
    out=profiles/r03_pmc_sq_${k}.txt; [ $k = pileup_rows ] && out=profiles/r03_pmc_sq_pol1m_default.txt
    { [ -f tools/r03_pmc_header_$k.txt ] && cat tools/r03_pmc_header_$k.txt
      for i in 1 2 3; do python3 tools/pmc_summary.py "$(cc pmc_default/pass$i)" $k; done; } > $out
  done
fi
ls profiles | grep -c r03_
