# one chunk of tools/final_bench.sh under its own timeout (a lost box costs one chunk, not the evidence run):   sh tools/final_bench_chunk.sh <tag> <chunk>
TAG=${1:-r06}; CH=$2
mkdir -p gpurun_out
case $CH in
fetch)
  timeout 500 python tools/collect_profiles.py $TAG > gpurun_out/collect_fetch.log 2>&1
  cp gpurun_out/pmc_${TAG}_hbm_traffic.json gpurun_out/pmc_${TAG}_sq_mix.json profiles/ 2>/dev/null
  timeout 300 python bench.py > gpurun_out/bench_${TAG}_fetch.json 2> gpurun_out/bench_${TAG}_fetch.err
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${TAG}_fetch_driver_args.json 2>/dev/null
  timeout 200 python bench.py --no-stagger --no-cpu-baseline > gpurun_out/bench_${TAG}_fetch_lockstep.json 2>/dev/null
  timeout 200 python bench.py --no-cpu-baseline --worlds-per-gpu 8192 > gpurun_out/bench_${TAG}_fetch_8192.json 2>/dev/null
  timeout 120 python tools/cost_probe.py > gpurun_out/cost_probe_${TAG}.txt 2>&1
  timeout 200 python tools/kernel_resources.py > gpurun_out/kernel_resources_${TAG}.txt 2>&1 ;;
pmc)   # counters of one other workload:  ... pmc <workload>
  export GRX_COLLECT_EXTRA="--preroll 10"
  timeout 400 python tools/collect_profiles.py $TAG pmc $3 > gpurun_out/collect_pmc_$3.log 2>&1
  timeout 400 python tools/collect_profiles.py $TAG sq $3 > gpurun_out/collect_sq_$3.log 2>&1
  unset GRX_COLLECT_EXTRA
  timeout 400 python tools/collect_profiles.py $TAG stats $3 > gpurun_out/collect_stats_$3.log 2>&1
  cp gpurun_out/pmc_${TAG}_hbm_traffic_$3.json gpurun_out/pmc_${TAG}_sq_mix_$3.json profiles/ 2>/dev/null
  timeout 400 python bench.py --workload $3 > gpurun_out/bench_${TAG}_$3.json 2> gpurun_out/bench_${TAG}_$3.err ;;
lines)   # bench lines without counters
  for w in mixed hand_reach adroit_door adroit_pen adroit_relocate; do
    timeout 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/bench_${TAG}_$w.json 2> gpurun_out/bench_${TAG}_$w.err
  done ;;
soak)
  timeout 900 python tools/soak.py 1000 > gpurun_out/soak_${TAG}.txt 2>&1 ;;
esac
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
rm -f gpurun_out/pmc_${TAG}_*.log gpurun_out/rocprof_${TAG}*.log
du -sh gpurun_out
