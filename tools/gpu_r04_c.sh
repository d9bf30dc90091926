#!/bin/sh
mkdir -p gpurun_out
L=gymnasium_robotics_amd/_lib
sh tools/ab_libs.sh "default $L/libgrx_hip_scan4inl.so $L/libgrx_hip_noscan4.so $L/libgrx_hip_nohull.so" --workload fetch --steps 60 --warmup 10 > gpurun_out/r04f_ab.txt 2>&1
cat gpurun_out/r04f_ab.txt | grep -v "^  File\|^    \|Traceback\|json" | cut -c1-200
python -m pytest tests/test_gpu_fetch.py -q -x 2>&1 | tail -3
