#!/bin/sh
# Fetch split step (GRX_FETCH_SPLIT = workgroups per world) on the default bench, one gpurun call:   sh tools/ab_fetch_split.sh [library] [worlds ...]
LIB=${1:-}; shift
[ -n "$LIB" ] && export GRX_HIP_LIB=$PWD/$LIB
OUT=gpurun_out/ab_r06_fetch_split.txt; : > $OUT
for n in ${@:-4096 8192 16384}; do
  for i in 1 2; do
    for p in 1 2 3 4 5; do
      GRX_FETCH_SPLIT=$p python bench.py --no-cpu-baseline --steps 60 --warmup 5 --no-sub-batches --no-north-star-share --no-long-window --worlds-per-gpu $n 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('split $p worlds $n ms_per_step %.3f kernel_ms %.3f value %.0f overflow %s' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value'], l['config'].get('capacity_overflow_worlds')))" >> $OUT
    done
  done
done
cat $OUT
