#!/bin/sh
# lane parameter sweep (margin / ttl / polling workgroups) on the workloads whose step ends with the overflow lane
mkdir -p gpurun_out
for w in adroit_door hand_touch kitchen; do
  sh tools/ab_libs.sh "default default:GRX_LANE_MARGIN=0.65 default:GRX_LANE_MARGIN=0.5 default:GRX_LANE_TTL=32 default:GRX_LANE_POLL=48 default:GRX_LANE_MARGIN=0.65+GRX_LANE_TTL=32+GRX_LANE_POLL=48" --workload $w --steps 60 --warmup 20 2>&1 | tee -a gpurun_out/r04_lane_sweep.txt
done
