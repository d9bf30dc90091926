# A/B/C of several builds of the HIP library on one bench workload inside ONE gpurun call (box-to-box variance is +-4 %):
#   sh tools/ab_libs.sh "<lib[:ENV=VAL+ENV=VAL...]> ..." [bench args]        lib = "default" or a path under the repo
SPECS=$1; shift
for i in 1 2; do
  for spec in $SPECS; do
    lib=${spec%%:*}; envs=""
    case "$spec" in *:*) envs=$(echo "${spec#*:}" | tr '+' ' ');; esac
    ( if [ "$lib" != default ]; then export GRX_HIP_LIB=$PWD/$lib; fi
      for e in $envs; do export "$e"; done
      python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$spec', ' '.join(sys.argv[1:]), 'ms_per_step %.3f kernel_ms %.3f value %.0f overflow %s' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value'], l['config'].get('capacity_overflow_worlds')))" "$@" )
  done
done
