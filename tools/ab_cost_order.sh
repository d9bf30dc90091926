#!/bin/sh
# cost-ordered dispatch for the kitchen and the Adroit families (GRX_KITCHEN_BALANCE / GRX_ADROIT_BALANCE = 0 / 1) on the default benches, one gpurun call
OUT=gpurun_out/ab_r06_cost_order.txt; : > $OUT
run() { python bench.py --no-cpu-baseline --workload $2 --steps $3 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$2 $1 value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT; }
for i in 1 2; do
  GRX_KITCHEN_BALANCE=0 run "order=off" kitchen 40
  GRX_KITCHEN_BALANCE=1 run "order=on" kitchen 40
done
for w in adroit adroit_door adroit_pen adroit_relocate; do
  GRX_ADROIT_BALANCE=0 run "order=off" $w 60
  GRX_ADROIT_BALANCE=1 run "order=on" $w 60
done
GRX_ADROIT_BALANCE=0 GRX_KITCHEN_BALANCE=0 run "order=off" mixed 60
GRX_ADROIT_BALANCE=1 GRX_KITCHEN_BALANCE=1 run "order=on" mixed 60
cat $OUT
