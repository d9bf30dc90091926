#!/bin/sh
# kitchen fast-kernel tables 192 / 2240 / 32 (5 worlds per CU) against 128 / 1280 / 24 (6 per CU; library: tools/build_variant.py kit128 "-DGRX_KITCHEN_CAP=128,1280,0,24" KITCHEN)
L=$PWD/gymnasium_robotics_amd/_lib
OUT=gpurun_out/ab_r05_kitchen_capacity.txt
mkdir -p gpurun_out; : > $OUT
for i in 1 2; do for spec in "default:" "kit128:128,1280,24"; do
  v=${spec%%:*}; cap=${spec#*:}
  if [ $v != default ]; then export GRX_HIP_LIB=$L/libgrx_hip_$v.so GRX_KITCHEN_CAP=$cap; else unset GRX_HIP_LIB GRX_KITCHEN_CAP; fi
  python bench.py --no-cpu-baseline --workload kitchen --steps 40 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$v $cap kitchen value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT
done; done
cat $OUT
