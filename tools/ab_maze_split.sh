#!/bin/sh
# maze split step (GRX_MAZE_SPLIT = workgroups per world) on the AntMaze bench (cfg 4), one gpurun call:   sh tools/ab_maze_split.sh [worlds ...]
OUT=gpurun_out/ab_r06_maze_split.txt; : > $OUT
WORLDS=${@:-8192 4096 16384}
for n in $WORLDS; do
  for i in 1 2; do
    for p in 1 2 3 5; do
      GRX_MAZE_SPLIT=$p python bench.py --no-cpu-baseline --workload antmaze --steps 100 --warmup 10 --no-sub-batches --worlds-per-gpu $n 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('split $p worlds $n ms_per_step %.3f kernel_ms %.3f value %.0f overflow %s' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value'], l['config'].get('capacity_overflow_worlds')))" >> $OUT
    done
  done
done
cat $OUT
