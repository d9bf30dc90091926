#!/bin/sh
# settle chains launched AHEAD of the step launch (GRX_CHAIN_FIRST=1) against behind it (0), default cfg 3 bench + chain device times, one gpurun call
OUT=gpurun_out/ab_r06_hand_chain_first.txt; : > $OUT
run() { timeout 300 python bench.py --no-cpu-baseline --workload $2 --steps 60 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$2 $1 value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT; }
for i in 1 2; do
  GRX_CHAIN_FIRST=0 run "chains behind the step launch" hand_touch
  GRX_CHAIN_FIRST=1 run "chains ahead of the step launch" hand_touch
done
p() { echo "== $1" >> $OUT; timeout 200 python tools/host_profile_hand.py 2>&1 | grep "^chain\|^step kernel" | awk '{ if ($1=="chain") printf "%s ", $12; else print }' >> $OUT; echo >> $OUT; }
GRX_CHAIN_FIRST=0 p "chain device times (ms), behind"
GRX_CHAIN_FIRST=1 p "chain device times (ms), ahead"
cat $OUT
