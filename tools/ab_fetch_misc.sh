#!/bin/sh
# with the split step on: moving-average weight of the cost order, order of the overlapped reset, tail order; default Fetch bench at 4096 / 8192, one gpurun call
OUT=gpurun_out/ab_r06_fetch_misc.txt; : > $OUT
run() { timeout 200 python bench.py --no-cpu-baseline --steps 60 --warmup 5 --no-sub-batches --no-north-star-share --no-long-window --worlds-per-gpu $2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$1 worlds $2 ms_per_step %.3f kernel_ms %.3f value %.0f' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value']))" >> $OUT; }
for n in 4096 8192; do
  for i in 1 2; do
    run "default" $n
    GRX_BALANCE_ALPHA=0.05 run "alpha0.05" $n
    GRX_BALANCE_ALPHA=0.2 run "alpha0.2" $n
    GRX_BALANCE_ALPHA=0.4 run "alpha0.4" $n
    GRX_FETCH_AHEAD_ORDER=before run "reset ahead queued before the step" $n
  done
done
cat $OUT
