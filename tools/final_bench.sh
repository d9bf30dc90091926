set -x
mkdir -p gpurun_out
for w in fetch hand_touch hand_reach antmaze adroit adroit_door adroit_pen adroit_relocate; do
  python bench.py --workload $w > gpurun_out/bench_r02_$w.json 2> gpurun_out/bench_r02_$w.err
  tail -c 300 gpurun_out/bench_r02_$w.json
done
python bench.py --no-stagger --no-cpu-baseline > gpurun_out/bench_r02_fetch_lockstep.json 2>/dev/null
python bench.py --workload hand_touch --no-stagger --no-cpu-baseline > gpurun_out/bench_r02_hand_touch_lockstep.json 2>/dev/null
