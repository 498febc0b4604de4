#!/bin/bash
# round 6, GPU call 23: eight waves on the small-problem GEMM's 64 x 128 tile (TIMHIP_GEMM_SMALL_W8=1) at 8 windows per GPU and on C1
TAG=${1:-r06aa}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
TIMHIP_GEMM_SMALL_W8=2 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm and not eight_phase and not pp_" > $OUT/pytest_gemm.log 2>&1
tail -3 $OUT/pytest_gemm.log
B="python bench.py --no-cpu-baseline --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 1 2; do
    TIMHIP_GEMM_SMALL_W8=$P timeout 600 $B 2>/dev/null | P=$P python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('small_w8', os.environ['P'], d['ms_per_step'], {k:(d[k].get('graph_replay') or {}).get('ms_per_step') for k in ('c2a_b8','c1','c4_train') if k in d})"
  done
done | tee $OUT/small_w8_ab.txt
