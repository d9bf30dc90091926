#!/bin/sh
# round 4: GPU tests + tolerance table on the current build, bench lines of every workload (bench.py defaults: 100 staggered steps)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r04_pytest.txt 2>&1; tail -8 gpurun_out/r04_pytest.txt
python tools/measure_tolerances.py > gpurun_out/r04_tolerances.txt 2>&1; grep -v "100.0 %" gpurun_out/r04_tolerances.txt | cut -c1-300
for w in fetch hand_touch hand_reach antmaze adroit adroit_door adroit_pen adroit_relocate kitchen mixed; do
  python bench.py --no-cpu-baseline --workload $w > gpurun_out/r04_bench_$w.json 2> gpurun_out/r04_bench_$w.err
  python -c "
import json,sys
l=json.loads(open('gpurun_out/r04_bench_$w.json').read().strip().splitlines()[-1]); print('$w', 'ms_per_step %.3f kernel_ms %.3f lane_ms %s value %.0f overflow %s' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['roofline'].get('kernel_plus_overflow_lane_ms'), l['value'], l['config'].get('capacity_overflow_worlds')))"
done
