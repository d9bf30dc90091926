#!/bin/sh
# Issue priority for the predicted stragglers (GRX_FETCH_PRIO = share of the worlds, GRX_FETCH_PRIO_LEVEL) on the default bench, one gpurun call:   sh tools/ab_fetch_prio.sh [worlds ...]
OUT=gpurun_out/ab_r06_fetch_prio.txt; : > $OUT
WORLDS=${@:-4096 8192}
for n in $WORLDS; do
  for i in 1 2; do
    for cfg in "0 3 2 1" "0.01 3 2 1" "0.03 3 2 1" "0.1 3 2 1" "0.25 3 2 1" "0.03 1 2 1" "0.03 3 1 1" "0.1 3 1 1" "0 3 2 0" "0.03 3 2 0"; do
      set -- $cfg
      GRX_FETCH_PRIO=$1 GRX_FETCH_PRIO_LEVEL=$2 GRX_FETCH_SPLIT=$3 GRX_TAIL_ORDER=$4 python bench.py --no-cpu-baseline --steps 60 --warmup 5 --no-sub-batches --no-north-star-share --no-long-window --worlds-per-gpu $n 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('prio $1 level $2 split $3 tail_order $4 worlds $n ms_per_step %.3f kernel_ms %.3f value %.0f overflow %s' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value'], l['config'].get('capacity_overflow_worlds')))" >> $OUT
    done
  done
done
cat $OUT
