#!/bin/sh
# settle chains pinned to the side stream that gets wave slots (GRX_CHAIN_PIN=1) against round-robin over the three streams (0), default cfg 3 bench, one gpurun call
OUT=gpurun_out/ab_r06_hand_chain_pin.txt; : > $OUT
run() { python bench.py --no-cpu-baseline --workload $2 --steps 60 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$2 $1 value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT; }
for i in 1 2; do
  GRX_CHAIN_PIN=0 run "round-robin" hand_touch
  GRX_CHAIN_PIN=1 run "pinned" hand_touch
done
GRX_CHAIN_PIN=1 GRX_CHAIN_STREAMS=4 run "pinned, 4 streams" hand_touch
cat $OUT
