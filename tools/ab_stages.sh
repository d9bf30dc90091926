#!/bin/sh
# out-of-phase sub-batches (bench.py --stages K) against the plain vector environment, same worlds per GPU, same timed region:   sh tools/ab_stages.sh
OUT=gpurun_out/ab_r05_stages.txt; : > $OUT
line='import json,sys; l=json.loads(sys.stdin.read()); r=l["roofline"]; print("%s value %.0f ms_per_step %.3f kernel_ms %.3f flagged %s" % (sys.argv[1], l["value"], l["ms_per_step"], r["kernel_ms"], l["config"].get("capacity_overflow_worlds")))'
for n in 4096 8192 16384; do for k in 1 2 1 2; do
  python bench.py --no-cpu-baseline --worlds-per-gpu $n --stages $k 2>/dev/null | python -c "$line" "fetch $n worlds stages=$k" >> $OUT
done; done
for w in antmaze adroit hand_touch; do for k in 1 2; do
  python bench.py --no-cpu-baseline --workload $w --stages $k 2>/dev/null | python -c "$line" "$w stages=$k" >> $OUT
done; done
cat $OUT
