#!/bin/sh
# overlapped (side-stream + commit) vs in-line same-step reset of the Fetch worlds, same box, alternating:   sh tools/ab_fetch_ahead_reset.sh
OUT=gpurun_out/ab_r05_fetch_ahead_reset.txt; : > $OUT
line='import json,sys; l=json.loads(sys.stdin.read()); r=l["roofline"]; print("%s value %.0f ms_per_step %.3f kernel_ms %.3f +rerun %s flagged %s" % (sys.argv[1], l["value"], l["ms_per_step"], r["kernel_ms"], r.get("kernel_plus_overflow_lane_ms"), l["config"].get("capacity_overflow_worlds")))'
for rep in 1 2 3; do for on in 1 0; do
  GRX_FETCH_AHEAD_RESET=$on python bench.py --no-cpu-baseline 2>/dev/null | python -c "$line" "ahead=$on default" >> $OUT
done
  GRX_FETCH_AHEAD_ORDER=after python bench.py --no-cpu-baseline 2>/dev/null | python -c "$line" "ahead=1(queued behind the step) default" >> $OUT
done
GRX_FETCH_AHEAD_ORDER=after python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "$line" "ahead=1(queued behind the step) steps20" >> $OUT
GRX_FETCH_AHEAD_ORDER=after python bench.py --no-cpu-baseline --no-stagger 2>/dev/null | python -c "$line" "ahead=1(queued behind the step) lockstep" >> $OUT
for on in 1 0; do
  GRX_FETCH_AHEAD_RESET=$on python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "$line" "ahead=$on steps20" >> $OUT
  GRX_FETCH_AHEAD_RESET=$on python bench.py --no-cpu-baseline --worlds-per-gpu 8192 2>/dev/null | python -c "$line" "ahead=$on 8192" >> $OUT
  GRX_FETCH_AHEAD_RESET=$on python bench.py --no-cpu-baseline --no-stagger 2>/dev/null | python -c "$line" "ahead=$on lockstep" >> $OUT
done
cat $OUT
