# bench lines of the out-of-phase sub-batch mode (bench.py --stages 2) + the families not yet measured with it:   sh tools/stages_evidence.sh   (ON the GPU box)
mkdir -p gpurun_out
python bench.py --stages 2 > gpurun_out/bench_r05_fetch_stages2.json 2>/dev/null
python bench.py --stages 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r05_fetch_stages2_driver_args.json 2>/dev/null
python bench.py --stages 2 --worlds-per-gpu 8192 --no-cpu-baseline > gpurun_out/bench_r05_fetch_8192_stages2.json 2>/dev/null
python bench.py --stages 2 --workload antmaze --no-cpu-baseline > gpurun_out/bench_r05_antmaze_stages2.json 2>/dev/null
python bench.py --stages 2 --workload adroit --no-cpu-baseline > gpurun_out/bench_r05_adroit_stages2.json 2>/dev/null
OUT=gpurun_out/ab_r05_stages_more.txt; : > $OUT
line='import json,sys; l=json.loads(sys.stdin.read()); r=l["roofline"]; print("%s value %.0f ms_per_step %.3f kernel_ms %.3f flagged %s" % (sys.argv[1], l["value"], l["ms_per_step"], r["kernel_ms"], l["config"].get("capacity_overflow_worlds")))'
for w in kitchen adroit_pen adroit_door adroit_relocate hand_reach; do for k in 1 2; do
  python bench.py --no-cpu-baseline --workload $w --stages $k 2>/dev/null | python -c "$line" "$w stages=$k" >> $OUT
done; done
for k in 3 4; do GPU_MAX_HW_QUEUES=16 python bench.py --no-cpu-baseline --workload antmaze --stages $k 2>/dev/null | python -c "$line" "antmaze stages=$k (GPU_MAX_HW_QUEUES=16)" >> $OUT; done
GPU_MAX_HW_QUEUES=16 python bench.py --no-cpu-baseline --workload adroit --stages 3 --worlds-per-gpu 16128 2>/dev/null | python -c "$line" "adroit 16128 worlds stages=3 (GPU_MAX_HW_QUEUES=16)" >> $OUT
cat $OUT
