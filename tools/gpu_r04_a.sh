#!/bin/sh
# round 4, GPU call A: GPU tests on the new build, tolerance table, A/B of the fp64 convex routine (inline / behind a call / fp32)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r04a_pytest.txt 2>&1; tail -25 gpurun_out/r04a_pytest.txt
python tools/measure_tolerances.py > gpurun_out/r04a_tolerances.txt 2>&1; tail -3 gpurun_out/r04a_tolerances.txt
LIBS="default gymnasium_robotics_amd/_lib/libgrx_hip_mprcall.so gymnasium_robotics_amd/_lib/libgrx_hip_mprf32.so"
for w in fetch hand_touch kitchen adroit_pen; do
  sh tools/ab_libs.sh "$LIBS" --workload $w --steps 40 --warmup 8 >> gpurun_out/r04a_ab.txt 2>&1
done
cat gpurun_out/r04a_ab.txt
