# GPU suite + the Adroit / mixed evidence files only (after an Adroit-only change):   sh tools/adroit_evidence.sh     (ON the GPU box, from the repo root)
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r05.txt 2>&1; tail -1 gpurun_out/gputest_r05.txt
export GRX_COLLECT_EXTRA="--preroll 10"
python tools/collect_profiles.py r05 pmc adroit > gpurun_out/collect_pmc.log 2>&1
python tools/collect_profiles.py r05 sq adroit > gpurun_out/collect_sq.log 2>&1
unset GRX_COLLECT_EXTRA
python tools/collect_profiles.py r05 stats adroit > gpurun_out/collect_stats.log 2>&1
cp gpurun_out/pmc_r05_hbm_traffic_adroit.json gpurun_out/pmc_r05_sq_mix_adroit.json profiles/
for w in adroit mixed; do python bench.py --workload $w > gpurun_out/bench_r05_$w.json 2> gpurun_out/bench_r05_$w.err; tail -c 200 gpurun_out/bench_r05_$w.json; done
for w in adroit_door adroit_pen adroit_relocate; do python bench.py --workload $w --no-cpu-baseline > gpurun_out/bench_r05_$w.json 2> gpurun_out/bench_r05_$w.err; done
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
