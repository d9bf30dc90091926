#!/bin/sh
# round 4, GPU call D: A/B of the hull tie-break (registers) and the gated object refinement: v2 = both, v2nohull = no tie-break, v2noobj = neither
mkdir -p gpurun_out
L=gymnasium_robotics_amd/_lib
sh tools/ab_libs.sh "$L/libgrx_hip_v2.so $L/libgrx_hip_v2nohull.so $L/libgrx_hip_v2noobj.so" --workload fetch --steps 60 --warmup 10 > gpurun_out/r04d_ab.txt 2>&1
sh tools/ab_libs.sh "$L/libgrx_hip_v2.so $L/libgrx_hip_v2nohull.so" --workload kitchen --steps 40 --warmup 8 >> gpurun_out/r04d_ab.txt 2>&1
cat gpurun_out/r04d_ab.txt
