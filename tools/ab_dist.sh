# the collective leg of bench.py on ONE rank (nccl group of size 1) against the run without it, in one gpurun call:  sh tools/ab_dist.sh [bench args]
p() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms_per_step %.3f kernel_ms %.3f value %.0f' % (l['ms_per_step'], l['roofline']['kernel_ms'], l['value']))"; }
for i in 1 2; do
python bench.py --no-cpu-baseline "$@" 2>/dev/null | p no_collective
GRX_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 1 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | p one_rank_nccl_gather
done
