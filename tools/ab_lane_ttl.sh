#!/bin/sh
# lane time-to-live for the other lane families (door, relocate, hand + touch), default benches, one gpurun call
OUT=gpurun_out/ab_r06_lane_ttl.txt; : > $OUT
run() { timeout 300 python bench.py --no-cpu-baseline --workload $2 --steps 60 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$2 $1 value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT; }
for w in adroit_door adroit_relocate hand_touch; do
  for i in 1 2; do
    run "default" $w
    GRX_LANE_TTL=1 run "ttl1" $w
    GRX_LANE_TTL=2 run "ttl2" $w
  done
done
timeout 300 python bench.py --no-cpu-baseline --workload kitchen --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('kitchen new default', l['value'], l['ms_per_step'])" >> $OUT
cat $OUT
