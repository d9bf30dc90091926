run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys,os
l=json.loads(sys.stdin.read()); print(' '.join(sys.argv[1:]), 'bal', os.environ.get('GRX_BENCH_BALANCE'), 'ms_per_step %.3f kernel_ms %.3f value %.0f' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value']))" "$@"; }
for b in 0 1 0 1; do GRX_BENCH_BALANCE=$b run --workload antmaze; done
