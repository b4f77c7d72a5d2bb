for args in "--workload pol_1m --positions 20" "--workload pol_1m --positions 5" "--workload pol_1m --positions 100" "--workload hiv_mix_200k --positions 20" "--workload hiv_mix_200k --positions 100" "--workload hiv_mix_200k --positions 300" "--workload amp20k_10m --reads 200000 --positions 20" "--workload amp20k_10m --reads 200000 --positions 300"; do
for m in 1 0; do
MSQ_PHASE_LOOKUP=$m timeout -k 10 200 python tools/bench_phase.py $args --steps 20 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lookup=$m', '$args', 'ms %.4f' % d['ms'])" || exit 1
done; done
