#!/bin/sh
# with the cost-ordered dispatch on: lane admission margin / time-to-live / polling workgroups / moving-average weight for the lane-bound families, one gpurun call
OUT=gpurun_out/ab_r06_cost_order_lane.txt; : > $OUT
run() { python bench.py --no-cpu-baseline --workload $2 --steps $3 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$2 $1 value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT; }
for w in adroit_door adroit_relocate; do
  run "default" $w 60
  GRX_LANE_MARGIN=0.9 run "margin0.9" $w 60
  GRX_LANE_MARGIN=0.95 run "margin0.95" $w 60
  GRX_LANE_MARGIN=0.9 GRX_LANE_TTL=2 run "margin0.9,ttl2" $w 60
  GRX_LANE_FIRST=0 run "lane_first0" $w 60
  GRX_BALANCE_ALPHA=0.3 run "alpha0.3" $w 60
done
run "default" kitchen 40
GRX_LANE_MARGIN=0.9 run "margin0.9" kitchen 40
GRX_LANE_POLL=48 run "poll48" kitchen 40
GRX_LANE_TTL=2 run "ttl2" kitchen 40
GRX_BALANCE_ALPHA=0.3 run "alpha0.3" kitchen 40
GRX_BALANCE_ALPHA=1.0 run "alpha1.0" kitchen 40
GRX_BALANCE_ALPHA=0.3 run "alpha0.3" adroit 60
cat $OUT
