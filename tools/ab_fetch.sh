# A/B of two builds of the HIP library on a bench workload inside ONE gpurun call (box-to-box variance is +-4 %):  sh tools/ab_fetch.sh <other.so> [bench args]
OTHER=$1; shift
for i in 1 2; do
  for lib in default "$OTHER"; do
    if [ "$lib" = default ]; then unset GRX_HIP_LIB; else export GRX_HIP_LIB=$PWD/$lib; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys,os
l=json.loads(sys.stdin.read()); print(os.environ.get('GRX_HIP_LIB','default').split('/')[-1], ' '.join(sys.argv[1:]), 'ms_per_step %.3f kernel_ms %.3f value %.0f' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value']))" "$@"
  done
done
