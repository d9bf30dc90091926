cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tl() { rm -rf /tmp/tl_$1; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$1 -o b -- python $R/bench.py --workload $1 --steps $2 --warmup 3 --no-cpu-baseline $5 > /dev/null 2>&1; python $R/tools/step_timeline.py /tmp/tl_$1 "$3" $4 > $R/gpurun_out/timeline_r06_$1.txt 2>&1; }
tl fetch 20 "grx_fetch_step_kernel<GrxShape<22" 0 "--no-sub-batches --no-north-star-share --no-long-window"
tl adroit_door 12 "grx_adroit_step_kernel" 3000 ""
tl kitchen 8 "grx_kitchen_step_kernel" 10000 ""
tl hand_touch 12 "grx_hand_step_kernel" 8000 ""
