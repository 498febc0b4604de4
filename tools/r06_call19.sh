#!/bin/bash
# round 6, GPU call 19: LDS stages of the small-problem GEMM instances (2 = before; 0 = by the block count; 3; 4): kernel tests, then
# the C2a step and the secondary configurations
TAG=${1:-r06v}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > $OUT/pytest_gemm.log 2>&1
tail -3 $OUT/pytest_gemm.log
B="python bench.py --no-cpu-baseline --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2; do
  for P in 2 0 3 4; do
    TIMHIP_GEMM_SMALL_NST=$P timeout 600 $B 2>/dev/null | P=$P python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('small_nst', os.environ['P'], d['ms_per_step'], {k:(d[k].get('graph_replay') or {}).get('ms_per_step') for k in ('c2a_b8','c2a_train','c2b','c1','c3','c4_train') if k in d})"
  done
done | tee $OUT/small_nst_ab.txt
