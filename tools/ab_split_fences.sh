#!/bin/sh
# the split steps' hand-off: write-through row + drained flag (the build) against agent-scope fences (libgrx_hip_agentfence.so = -DGRX_SPLIT_AGENT_FENCES), parts 1 - 5, one gpurun call
OUT=gpurun_out/ab_r06_split_fences.txt; : > $OUT
OLD=$PWD/gymnasium_robotics_amd/_lib/libgrx_hip_agentfence.so
fetch() { timeout 200 python bench.py --no-cpu-baseline --steps 60 --warmup 5 --no-sub-batches --no-north-star-share --no-long-window --worlds-per-gpu $2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('fetch $1 worlds $2 ms_per_step %.3f kernel_ms %.3f value %.0f' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value']))" >> $OUT; }
ant() { timeout 200 python bench.py --no-cpu-baseline --workload antmaze --steps 100 --warmup 10 --no-sub-batches --worlds-per-gpu $2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('antmaze $1 worlds $2 ms_per_step %.3f kernel_ms %.3f value %.0f' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value']))" >> $OUT; }
for n in 4096 8192 16384; do
  for p in 1 2 3 4 5; do
    GRX_FETCH_SPLIT=$p fetch "write-through split $p" $n
  done
  GRX_HIP_LIB=$OLD GRX_FETCH_SPLIT=2 fetch "agent-fences split 2" $n
  GRX_HIP_LIB=$OLD GRX_FETCH_SPLIT=4 fetch "agent-fences split 4" $n
done
for n in 8192 4096 16384; do
  for p in 1 2 3 5; do
    GRX_MAZE_SPLIT=$p ant "write-through split $p" $n
  done
  GRX_HIP_LIB=$OLD GRX_MAZE_SPLIT=2 ant "agent-fences split 2" $n
  GRX_HIP_LIB=$OLD GRX_MAZE_SPLIT=5 ant "agent-fences split 5" $n
done
cat $OUT
