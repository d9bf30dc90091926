#!/bin/sh
# polling workgroups of the standing overflow lane, per lane-mode family:   sh tools/ab_lane_poll.sh
OUT=gpurun_out/ab_r05_lane_poll.txt; : > $OUT
for w in adroit_door adroit_relocate hand_touch; do for p in 16 32 48; do
  GRX_LANE_POLL=$p python bench.py --no-cpu-baseline --workload $w 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$w poll $p value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT
done; done
cat $OUT
