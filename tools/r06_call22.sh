#!/bin/bash
# round 6, GPU call 22: 128-deep stages in the small-problem GEMM (TIMHIP_GEMM_SMALL_BK=128) at 8 windows per GPU and on C1
TAG=${1:-r06z}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
TIMHIP_GEMM_SMALL_BK=128 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm and not eight_phase and not pp_" > $OUT/pytest_gemm.log 2>&1
tail -3 $OUT/pytest_gemm.log
B="python bench.py --no-cpu-baseline --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 64 128; do
    TIMHIP_GEMM_SMALL_BK=$P timeout 600 $B 2>/dev/null | P=$P python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('small_bk', os.environ['P'], d['ms_per_step'], {k:(d[k].get('graph_replay') or {}).get('ms_per_step') for k in ('c2a_b8','c1','c4_train') if k in d})"
  done
done | tee $OUT/small_bk_ab.txt
