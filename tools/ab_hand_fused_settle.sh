#!/bin/sh
# settle chains of the overlapped hand-manipulation reset: one repeat launch (GRX_HAND_FUSED_SETTLE=1) against ten launches (0), default cfg 3 bench, one gpurun call
OUT=gpurun_out/ab_r06_hand_fused_settle.txt; : > $OUT
run() { python bench.py --no-cpu-baseline --workload $2 --steps 60 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$2 $1 value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT; }
for i in 1 2; do
  GRX_HAND_FUSED_SETTLE=0 run "ten launches" hand_touch
  GRX_HAND_FUSED_SETTLE=1 run "one repeat launch" hand_touch
done
GRX_HAND_FUSED_SETTLE=1 GRX_CHAIN_LOOKAHEAD=1 run "one repeat launch, lookahead 1" hand_touch
cat $OUT
