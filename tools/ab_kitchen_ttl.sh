#!/bin/sh
# kitchen, cost-ordered dispatch on: the lane's time-to-live / admission margin (GRX_LANE_TTL, GRX_LANE_MARGIN), default bench, one gpurun call
OUT=gpurun_out/ab_r06_kitchen_ttl_b.txt; : > $OUT
run() { timeout 300 python bench.py --no-cpu-baseline --workload kitchen --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('kitchen $1 value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT; }
for i in 1 2; do
  GRX_LANE_TTL=1 run "ttl1"
  GRX_LANE_TTL=0 run "ttl0"
  GRX_LANE_TTL=1 GRX_LANE_MARGIN=0.9 run "ttl1,margin0.9"
  GRX_LANE_TTL=1 GRX_LANE_MARGIN=0.7 run "ttl1,margin0.7"
  GRX_LANE_TTL=1 GRX_LANE_POLL=24 run "ttl1,poll24"
done
cat $OUT
