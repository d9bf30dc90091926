#!/bin/sh
# kitchen: lane admission margin x time-to-live x polling workgroups on the default bench (one gpurun call)
OUT=gpurun_out/ab_r06_kitchen_lane_margin.txt; : > $OUT
run() { python bench.py --no-cpu-baseline --workload kitchen --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$1 value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT; }
run "margin0.8,ttl4,poll32(default)"
GRX_LANE_MARGIN=0.65 run "margin0.65,ttl4"
GRX_LANE_MARGIN=0.5 run "margin0.5,ttl4"
GRX_LANE_MARGIN=0.65 GRX_LANE_TTL=8 run "margin0.65,ttl8"
GRX_LANE_MARGIN=0.5 GRX_LANE_TTL=8 run "margin0.5,ttl8"
GRX_LANE_MARGIN=0.5 GRX_LANE_TTL=16 run "margin0.5,ttl16"
GRX_LANE_MARGIN=0.65 GRX_LANE_POLL=64 run "margin0.65,ttl4,poll64"
run "margin0.8,ttl4,poll32(default again)"
cat $OUT
