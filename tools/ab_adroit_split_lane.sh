OUT=gpurun_out/ab_r06_adroit_split.txt; : > $OUT
run() { timeout 300 python bench.py --no-cpu-baseline --workload $2 --steps 60 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); r=l.get('roofline') or {}; print('$2 $1 value %.0f ms_per_step %.3f kernel_ms %s +lane %s overflow %s' % (l['value'], l['ms_per_step'], r.get('kernel_ms'), r.get('kernel_plus_overflow_lane_ms'), l['config'].get('capacity_overflow_worlds')))" >> $OUT; }
for w in adroit_relocate adroit_door; do
  for p in 1 2 3 5; do
    GRX_ADROIT_SPLIT=$p run "split $p" $w
  done
done
GRX_ADROIT_SPLIT=5 run "split 5" mixed
cat $OUT
