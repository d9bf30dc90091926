# round-end evidence run (ON the GPU box, from the repo root): every bench line, rocprofv3 kernel stats, PMC traffic and the SQ issue mix of the BASELINE configs
#   sh tools/final_bench.sh [tag]        (default tag: r05)
TAG=${1:-r05}
set -x
mkdir -p gpurun_out
python tools/collect_profiles.py $TAG > gpurun_out/collect_fetch.log 2>&1                     # cfg 2: kernel stats + HBM traffic + SQ mix
export GRX_COLLECT_EXTRA="--preroll 10"                                                      # (the counter passes of the long-horizon workloads: ten steps in, not a whole episode)
python tools/collect_profiles.py $TAG pmc antmaze hand_touch adroit kitchen > gpurun_out/collect_pmc.log 2>&1
python tools/collect_profiles.py $TAG sq antmaze hand_touch adroit kitchen > gpurun_out/collect_sq.log 2>&1
unset GRX_COLLECT_EXTRA
python tools/collect_profiles.py $TAG stats antmaze hand_touch adroit kitchen > gpurun_out/collect_stats.log 2>&1
cp gpurun_out/pmc_${TAG}_hbm_traffic*.json gpurun_out/pmc_${TAG}_sq_mix*.json profiles/   # the bench lines below quote the traffic / issue mix measured in THIS run
for w in fetch hand_touch antmaze adroit kitchen mixed; do
  python bench.py --workload $w > gpurun_out/bench_${TAG}_$w.json 2> gpurun_out/bench_${TAG}_$w.err
  tail -c 200 gpurun_out/bench_${TAG}_$w.json
done
for w in hand_reach adroit_door adroit_pen adroit_relocate; do
  python bench.py --workload $w --no-cpu-baseline > gpurun_out/bench_${TAG}_$w.json 2> gpurun_out/bench_${TAG}_$w.err
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${TAG}_fetch_driver_args.json 2>/dev/null
python bench.py --no-stagger --no-cpu-baseline > gpurun_out/bench_${TAG}_fetch_lockstep.json 2>/dev/null
python bench.py --no-cpu-baseline --worlds-per-gpu 8192 > gpurun_out/bench_${TAG}_fetch_8192.json 2>/dev/null
python tools/cost_probe.py > gpurun_out/cost_probe_${TAG}.txt 2>&1
python tools/soak.py 1000 > gpurun_out/soak_${TAG}.txt 2>&1
python tools/kernel_resources.py > gpurun_out/kernel_resources_${TAG}.txt 2>&1
# the raw rocprofv3 output directories are tens of MB each: only the summaries travel back (gpurun_out/ is capped at 64 MiB)
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
rm -f gpurun_out/*.log
du -sh gpurun_out
