#!/bin/sh
# HBM traffic (two PMC passes) and step time of several builds of the HIP library on one bench workload, inside ONE gpurun call:
#   sh tools/ab_traffic.sh "<variant> ..." <workload> [tag]        variant = "default" or the <name> of gymnasium_robotics_amd/_lib/libgrx_hip_<name>.so
L=$PWD/gymnasium_robotics_amd/_lib
W=$2; TAG=${3:-ab}
OUT=gpurun_out/${TAG}_${W}_traffic.txt
mkdir -p gpurun_out; : > $OUT
SUF=_$W; [ "$W" = fetch ] && SUF=""
for v in $1; do
  if [ $v != default ]; then export GRX_HIP_LIB=$L/libgrx_hip_$v.so; else unset GRX_HIP_LIB; fi
  GRX_COLLECT_EXTRA="--preroll 0" python tools/collect_profiles.py ab_$v pmc $W > /dev/null 2>&1
  echo "== $v" >> $OUT
  sed -n 2p gpurun_out/pmc_ab_${v}_hbm_traffic$SUF.txt | cut -c1-400 >> $OUT
  tail -3 gpurun_out/pmc_ab_${v}_hbm_traffic$SUF.txt >> $OUT
  python bench.py --no-cpu-baseline --workload $W --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('bench value %.0f ms_per_step %.3f kernel_ms %.3f overflow %s' % (l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['config'].get('capacity_overflow_worlds')))" >> $OUT
done
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $OUT
