#!/bin/sh
# kitchen: HBM traffic and step time of builds that differ in the convex routine's register footprint
L=$PWD/gymnasium_robotics_amd/_lib
mkdir -p gpurun_out
for v in default wf mprf32; do
  if [ $v != default ]; then export GRX_HIP_LIB=$L/libgrx_hip_$v.so; else unset GRX_HIP_LIB; fi
  python tools/collect_profiles.py ab_$v pmc kitchen > /dev/null 2>&1
  echo "== $v" >> gpurun_out/r04_kitchen_traffic_ab.txt
  tail -3 gpurun_out/pmc_ab_${v}_hbm_traffic_kitchen.txt >> gpurun_out/r04_kitchen_traffic_ab.txt
  python bench.py --no-cpu-baseline --workload kitchen --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('bench', l['value'], l['ms_per_step'], l['roofline']['kernel_ms'])" >> gpurun_out/r04_kitchen_traffic_ab.txt
done
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat gpurun_out/r04_kitchen_traffic_ab.txt
