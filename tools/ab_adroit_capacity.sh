#!/bin/sh
# Adroit fast-kernel table capacities (rows, pool words, contacts) -> worlds per CU -> throughput, with the overflow lane taking the worlds that exceed them:
#   sh tools/ab_adroit_capacity.sh        (libraries: tools/build_variant.py adr112 / adr96 with -DGRX_ADROIT_ME / _JP / _MC)
L=$PWD/gymnasium_robotics_amd/_lib
OUT=gpurun_out/ab_r05_adroit_capacity.txt
mkdir -p gpurun_out; : > $OUT
for spec in "default:" "adr112:112,1280,24" "adr96:96,1024,24"; do
  v=${spec%%:*}; cap=${spec#*:}
  if [ $v != default ]; then export GRX_HIP_LIB=$L/libgrx_hip_$v.so GRX_ADROIT_CAP=$cap; else unset GRX_HIP_LIB GRX_ADROIT_CAP; fi
  for w in adroit adroit_door adroit_pen adroit_relocate; do
    python bench.py --no-cpu-baseline --workload $w --steps 60 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$v $cap $w value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s overflow %s' % (l['value'], l['ms_per_step'], r['kernel_ms'], r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT
  done
done
cat $OUT
