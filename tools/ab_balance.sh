run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys,os
l=json.loads(sys.stdin.read()); print(' '.join(sys.argv[1:]), 'alpha', os.environ.get('GRX_BALANCE_ALPHA'), 'bal', os.environ.get('GRX_BENCH_BALANCE'), 'ms_per_step %.3f kernel_ms %.3f value %.0f' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value']))" "$@"; }
for a in 0.1 0.25 0.5 1.0 0.1 0.25; do GRX_BALANCE_ALPHA=$a run; done
GRX_BALANCE_ALPHA=0.1 run --worlds-per-gpu 8192
for b in 0 1 0 1; do GRX_BENCH_BALANCE=$b run --workload hand_touch --steps 40; done
GRX_BENCH_BALANCE=1 run --workload hand_reach --steps 40; GRX_BENCH_BALANCE=0 run --workload hand_reach --steps 40
