L=$PWD/gymnasium_robotics_amd/_lib
run() { python bench.py --no-cpu-baseline --workload adroit 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$1 value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))"; }
run "hammer96/1024/24,entry"
GRX_LANE_MODE=lane run "hammer96/1024/24,lane"
GRX_LANE_MODE=lane GRX_LANE_POLL=32 run "hammer96/1024/24,lane,poll32"
GRX_HIP_LIB=$L/libgrx_hip_adrA.so GRX_ADROIT_CAP=64,768,24 GRX_LANE_MODE=lane run "hammer64/768/24,lane"
GRX_HIP_LIB=$L/libgrx_hip_adrA.so GRX_ADROIT_CAP=64,768,24 GRX_LANE_MODE=lane GRX_LANE_POLL=32 run "hammer64/768/24,lane,poll32"
