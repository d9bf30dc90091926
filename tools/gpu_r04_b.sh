#!/bin/sh
# round 4, GPU call B: GPU tests + tolerance table on the parity build, bench lines of the BASELINE workloads
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r04b_pytest.txt 2>&1; tail -30 gpurun_out/r04b_pytest.txt
python tools/measure_tolerances.py > gpurun_out/r04b_tolerances.txt 2>&1; grep -v "100.0 %" gpurun_out/r04b_tolerances.txt | cut -c1-260
for w in fetch hand_touch kitchen adroit adroit_door adroit_relocate mixed; do
  python bench.py --no-cpu-baseline --workload $w > gpurun_out/r04b_bench_$w.json 2> gpurun_out/r04b_bench_$w.err
  python -c "
import json,sys
l=json.loads(open('gpurun_out/r04b_bench_$w.json').read().strip().splitlines()[-1]); print('$w', 'ms_per_step %.3f kernel_ms %.3f value %.0f overflow %s' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value'], l['config'].get('capacity_overflow_worlds')))"
done
