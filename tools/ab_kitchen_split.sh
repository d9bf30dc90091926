#!/bin/sh
# kitchen split step (GRX_KITCHEN_SPLIT = workgroups per world) on the default bench and the cfg 5 batch, one gpurun call
OUT=gpurun_out/ab_r06_kitchen_split.txt; : > $OUT
run() { timeout 300 python bench.py --no-cpu-baseline --workload $2 --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l.get('roofline') or {}; print('$2 $1 value %.0f ms_per_step %.3f kernel_ms %s +lane %s overflow %s' % (l['value'], l['ms_per_step'], r.get('kernel_ms'), r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT; }
for p in 1 2 4 5 8; do
  GRX_KITCHEN_SPLIT=$p run "split $p" kitchen
done
GRX_KITCHEN_SPLIT=1 run "split 1" mixed
GRX_KITCHEN_SPLIT=4 run "split 4" mixed
cat $OUT
