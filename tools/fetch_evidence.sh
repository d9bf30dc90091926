set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r05.txt 2>&1; tail -1 gpurun_out/gputest_r05.txt
python tools/collect_profiles.py r05 > gpurun_out/collect_fetch.log 2>&1
cp gpurun_out/pmc_r05_hbm_traffic.json gpurun_out/pmc_r05_sq_mix.json profiles/
python bench.py --workload fetch > gpurun_out/bench_r05_fetch.json 2> gpurun_out/bench_r05_fetch.err; tail -c 300 gpurun_out/bench_r05_fetch.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r05_fetch_driver_args.json 2>/dev/null
python bench.py --no-stagger --no-cpu-baseline > gpurun_out/bench_r05_fetch_lockstep.json 2>/dev/null
python bench.py --no-cpu-baseline --worlds-per-gpu 8192 > gpurun_out/bench_r05_fetch_8192.json 2>/dev/null
python bench.py --no-cpu-baseline --preroll 1500 > gpurun_out/bench_r05_fetch_preroll1500.json 2>/dev/null
python tools/cost_probe.py > gpurun_out/cost_probe_r05.txt 2>&1
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
tail -5 gpurun_out/collect_fetch.log
