# round-end evidence run (ON the GPU box, from the repo root): every bench line with its CPU baseline, rocprofv3 kernel stats + PMC traffic of every workload
#   sh tools/final_bench.sh [tag]        (default tag: r04)
TAG=${1:-r04}
set -x
mkdir -p gpurun_out
python tools/collect_profiles.py $TAG > gpurun_out/collect_fetch.log 2>&1
python tools/collect_profiles.py $TAG workloads antmaze hand_touch hand_reach adroit adroit_door adroit_pen adroit_relocate kitchen > gpurun_out/collect_workloads.log 2>&1
cp gpurun_out/pmc_${TAG}_hbm_traffic*.json profiles/   # the bench lines below quote the traffic measured in THIS run
for w in fetch hand_touch hand_reach antmaze adroit adroit_door adroit_pen adroit_relocate kitchen mixed; do
  python bench.py --workload $w > gpurun_out/bench_${TAG}_$w.json 2> gpurun_out/bench_${TAG}_$w.err
  tail -c 200 gpurun_out/bench_${TAG}_$w.json
done
python bench.py --no-stagger --no-cpu-baseline > gpurun_out/bench_${TAG}_fetch_lockstep.json 2>/dev/null
python bench.py --no-cpu-baseline --worlds-per-gpu 8192 > gpurun_out/bench_${TAG}_fetch_8192.json 2>/dev/null
python tools/cost_probe.py > gpurun_out/cost_probe_${TAG}.txt 2>&1
python tools/soak.py 1000 > gpurun_out/soak_${TAG}.txt 2>&1
# the raw rocprofv3 output directories are tens of MB each: only the summaries travel back (gpurun_out/ is capped at 64 MiB)
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
rm -f gpurun_out/*.log
du -sh gpurun_out
