set -e
L=gymnasium_robotics_amd/_lib
cp $L/libgrx_hip.so $L/alt_default.so
for rep in 1 2; do
for v in default maxilp bias0 bias50 trk; do
  cp $L/alt_$v.so $L/libgrx_hip.so
  for w in fetch hand_touch antmaze; do
    if [ $w = fetch ]; then A=""; else A="--workload $w"; fi
    r=$(python bench.py --steps 100 --warmup 10 --no-cpu-baseline $A 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f'%d['value'])")
    echo "$rep $v $w $r"
  done
done
done
