#!/bin/sh
L=gymnasium_robotics_amd/_lib
sh tools/ab_libs.sh "$L/libgrx_hip_med3.so $L/libgrx_hip_nosecond.so $L/libgrx_hip_nohull.so" --workload fetch --steps 60 --warmup 10 2>&1 | tee -a gpurun_out/r04i_ab.txt
GRX_HIP_LIB=$PWD/$L/libgrx_hip_med3.so python -m pytest tests/test_gpu_anchors.py tests/test_gpu_fetch.py -q --tb=short 2>&1 | grep -v "^$" | tail -5
