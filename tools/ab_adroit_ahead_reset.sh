#!/bin/sh
# overlapped (side stream + commit) vs in-line same-step reset of the Adroit families whose draws are made on the device:   sh tools/ab_adroit_ahead_reset.sh
OUT=gpurun_out/ab_r05_adroit_ahead_reset.txt; : > $OUT
line='import json,sys; l=json.loads(sys.stdin.read()); r=l["roofline"]; print("%s value %.0f ms_per_step %.3f kernel_ms %.3f +lane %s flagged %s" % (sys.argv[1], l["value"], l["ms_per_step"], r["kernel_ms"], r.get("kernel_plus_overflow_lane_ms"), l["config"].get("capacity_overflow_worlds")))'
for w in adroit adroit_door adroit_relocate; do for on in 1 0 1 0; do
  GRX_ADROIT_AHEAD_RESET=$on python bench.py --no-cpu-baseline --workload $w 2>/dev/null | python -c "$line" "$w ahead=$on" >> $OUT
done; done
cat $OUT
