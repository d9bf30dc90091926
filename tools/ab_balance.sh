# A/B of the cost-ordered dispatch for the hand families (off by default there) inside ONE gpurun call:  sh tools/ab_balance.sh [hand_touch|hand_reach]
W=${1:-hand_touch}
run() { python bench.py --no-cpu-baseline --workload $W --steps 40 2>/dev/null | python -c "
import json,sys,os
l=json.loads(sys.stdin.read()); print('$W', 'balance', os.environ.get('GRX_BENCH_BALANCE'), 'ms_per_step %.3f kernel_ms %.3f value %.0f' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value']))"; }
for b in 0 1 0 1; do GRX_BENCH_BALANCE=$b run; done
