#!/bin/sh
# hand split step (GRX_HAND_SPLIT = workgroups per world) on the cfg 3 and HandReach benches, one gpurun call
OUT=gpurun_out/ab_r06_hand_split.txt; : > $OUT
run() { timeout 300 python bench.py --no-cpu-baseline --workload $2 --steps 60 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l.get('roofline') or {}; print('$2 $1 value %.0f ms_per_step %.3f kernel_ms %s +lane %s overflow %s' % (l['value'], l['ms_per_step'], r.get('kernel_ms'), r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT; }
for w in hand_touch hand_reach; do
  for p in 1 2 4 5 1 4; do
    GRX_HAND_SPLIT=$p run "split $p" $w
  done
done
cat $OUT
