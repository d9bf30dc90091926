#!/bin/sh
# second pass over the hammer / pen fast tables: 8 worlds per CU
L=$PWD/gymnasium_robotics_amd/_lib
OUT=gpurun_out/ab_r05_adroit_capacity2.txt; : > $OUT
run() { python bench.py --no-cpu-baseline --workload $2 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$1 $2 value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT; }
run "hammer96/1024/24(7perCU)" adroit
run "pen112/1280/24(7perCU)" adroit_pen
GRX_HIP_LIB=$L/libgrx_hip_adrA.so GRX_ADROIT_CAP=64,768,24 run "hammer64/768/24(8perCU)" adroit
GRX_HIP_LIB=$L/libgrx_hip_adrA.so GRX_ADROIT_CAP=80,1024,24 run "pen80/1024/24(8perCU)" adroit_pen
GRX_HIP_LIB=$L/libgrx_hip_adrB.so GRX_ADROIT_CAP=80,768,16 run "hammer80/768/16(8perCU)" adroit
GRX_HIP_LIB=$L/libgrx_hip_adrB.so GRX_ADROIT_CAP=80,896,24 run "pen80/896/24(8perCU)" adroit_pen
cat $OUT
