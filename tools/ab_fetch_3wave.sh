# Upper bound of a FetchPickAndPlace FAST kernel that carries no hull-pair routine (168 VGPRs, 3 waves per SIMD) on smaller tables (96 rows / 1 024 pool words / 24 contacts: 10 worlds per CU),
# with the overflow re-runs switched off: what a mid-step hand-off of the hull worlds could reach AT MOST (the hull worlds are simply not collided here -- wrong physics, timing only).
#   build: python -c "import __graft_entry__ as g; g.build_hip(out='gymnasium_robotics_amd/_lib/libgrx_x3.so', extra_flags=['-DGRX_FETCH_ME=96','-DGRX_FETCH_JP=1024','-DGRX_FETCH_MC=24','-DGRX_FETCH_PICK_FLAGS=0','-DGRX_MATCH_ANY_MESH=1','-DGRX_FETCH_WAVES(S)=3'])"
#   run (GPU box): sh tools/ab_fetch_3wave.sh
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys,os
l=json.loads(sys.stdin.read()); print(os.environ.get('GRX_HIP_LIB','default').split('/')[-1], os.environ.get('GRX_FETCH_CAP',''), ' '.join(sys.argv[1:]), 'ms_per_step %.3f kernel_ms %.3f value %.0f flagged %d' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value'], l['config']['capacity_overflow_worlds']))" "$@"; }
for i in 1 2; do
  for n in 4096 8192 16384; do
    unset GRX_HIP_LIB GRX_FETCH_CAP GRX_NO_OVERFLOW_RERUN
    run --worlds-per-gpu $n --steps 60
    export GRX_HIP_LIB=$PWD/gymnasium_robotics_amd/_lib/libgrx_x3.so GRX_FETCH_CAP=96,1024,24 GRX_NO_OVERFLOW_RERUN=1
    run --worlds-per-gpu $n --steps 60
  done
done
