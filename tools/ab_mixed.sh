L=$PWD/gymnasium_robotics_amd/_lib
run() { python bench.py --no-cpu-baseline --workload mixed 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$1 value %.0f ms_per_step %.3f kernel_ms %s overflow %s' % (l['value'], l['ms_per_step'], l['config'].get('kernel_ms'), l['config'].get('capacity_overflow_worlds')))"; }
run "r05-default"
GRX_LANE_POLL=16 run "poll16"
GRX_HIP_LIB=$L/libgrx_hip_kit192.so GRX_KITCHEN_CAP=192,2240,32 GRX_LANE_POLL=16 run "kitchen192+poll16"
GRX_HIP_LIB=$L/libgrx_hip_adr144.so GRX_ADROIT_CAP=144,2032,32 run "hammer144"
