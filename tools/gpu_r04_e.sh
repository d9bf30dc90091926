#!/bin/sh
# lane parameter sweep, the other direction (fewer worlds in the lane) + the model-sharing test
mkdir -p gpurun_out
python -m pytest tests/test_gpu_api.py -q -x -m gpu 2>&1 | tail -3
for w in adroit_door hand_touch; do
  sh tools/ab_libs.sh "default default:GRX_LANE_MARGIN=0.9 default:GRX_LANE_TTL=4 default:GRX_LANE_MARGIN=0.9+GRX_LANE_TTL=4 default:GRX_LANE_TTL=2" --workload $w --steps 60 --warmup 20 2>&1 | tee -a gpurun_out/r04_lane_sweep2.txt
done
