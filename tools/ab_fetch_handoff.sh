# A/B of the FetchPickAndPlace hand-off build's launch parameters inside ONE gpurun call:  sh tools/ab_fetch_handoff.sh
run() { python bench.py --no-cpu-baseline --steps 60 "$@" 2>/dev/null | python -c "
import json,sys,os
l=json.loads(sys.stdin.read()); r=l['roofline']; print(' '.join(f'{k}={os.environ[k]}' for k in ('GRX_FETCH_HANDOFF','GRX_LANE_FIRST','GRX_LANE_SPACER','GRX_FETCH_TTL','GRX_FETCH_POLL','GRX_LANE_MARGIN') if k in os.environ), ' '.join(sys.argv[1:]), '| ms_per_step %.3f fast kernel %.3f group %.3f value %.0f flagged %d' % (l['ms_per_step'], r['kernel_ms'], r['kernel_plus_overflow_lane_ms'], l['value'], l['config']['capacity_overflow_worlds']))" "$@"; }
export GRX_LANE_FIRST=1 GRX_LANE_SPACER=50000
for n in 4096 8192 16384; do
  GRX_FETCH_HANDOFF=0 run --worlds-per-gpu $n
  GRX_FETCH_HANDOFF=1 run --worlds-per-gpu $n
  GRX_FETCH_HANDOFF=1 GRX_FETCH_TTL=1 GRX_LANE_MARGIN=1.0 run --worlds-per-gpu $n
  GRX_FETCH_HANDOFF=1 GRX_FETCH_TTL=1 GRX_LANE_MARGIN=1.0 GRX_FETCH_POLL=64 run --worlds-per-gpu $n
  GRX_FETCH_HANDOFF=1 GRX_FETCH_TTL=2 GRX_LANE_MARGIN=1.0 GRX_FETCH_POLL=32 run --worlds-per-gpu $n
done
