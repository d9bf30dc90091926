line='import json,sys; l=json.loads(sys.stdin.read()); r=l["roofline"]; print("%s value %.0f ms_per_step %.3f kernel_ms %.3f flagged %s" % (sys.argv[1], l["value"], l["ms_per_step"], r["kernel_ms"], l["config"].get("capacity_overflow_worlds")))'
OUT=gpurun_out/ab_r05_hw_queues.txt; : > $OUT
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --workload hand_touch 2>/dev/null | python -c "$line" "hand_touch plain GPU_MAX_HW_QUEUES=$q" >> $OUT
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --workload hand_touch --stages 2 2>/dev/null | python -c "$line" "hand_touch stages=2 GPU_MAX_HW_QUEUES=$q" >> $OUT
done
for q in 8 16; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline 2>/dev/null | python -c "$line" "fetch plain GPU_MAX_HW_QUEUES=$q" >> $OUT
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --stages 2 2>/dev/null | python -c "$line" "fetch stages=2 GPU_MAX_HW_QUEUES=$q" >> $OUT
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --workload kitchen 2>/dev/null | python -c "$line" "kitchen plain GPU_MAX_HW_QUEUES=$q" >> $OUT
done
GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline --workload mixed 2>/dev/null | python -c "$line" "mixed GPU_MAX_HW_QUEUES=8" >> $OUT
cat $OUT
