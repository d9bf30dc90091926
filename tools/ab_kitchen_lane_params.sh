#!/bin/sh
# kitchen: fast-kernel tables x overflow-lane parameters on the default bench (one whole episode of pre-roll, 100 timed steps)
#   sh tools/ab_kitchen_lane_params.sh       (library of the old tables: tools/build_variant.py kit192 "-DGRX_KITCHEN_CAP=192,2240,0,32" KITCHEN)
OUT=gpurun_out/ab_r05_kitchen_lane_params.txt; : > $OUT
run() { python bench.py --no-cpu-baseline --workload kitchen 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$1 value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT; }
GRX_HIP_LIB=$PWD/gymnasium_robotics_amd/_lib/libgrx_hip_kit192.so GRX_KITCHEN_CAP=192,2240,32 run "tables192/2240/32(r04),poll16"
GRX_HIP_LIB=$PWD/gymnasium_robotics_amd/_lib/libgrx_hip_kit192.so GRX_KITCHEN_CAP=192,2240,32 GRX_LANE_POLL=48 run "tables192/2240/32(r04),poll48"
run "tables128/1280/24,poll16"
GRX_LANE_POLL=32 run "tables128/1280/24,poll32"
GRX_LANE_POLL=48 run "tables128/1280/24,poll48"
GRX_LANE_POLL=96 run "tables128/1280/24,poll96"
GRX_LANE_POLL=48 GRX_LANE_TTL=2 run "tables128/1280/24,poll48,ttl2"
cat $OUT
