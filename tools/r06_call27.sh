#!/bin/bash
# round 6, GPU call 27: ragged contraction lengths in the eight-phase weight-gradient kernel (C4: M = 7984): kernel tests, detection
# training parity/graph tests, c4_train with the paired launch against layer by layer
TAG=${1:-r06aj}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad_group_eight_phase" > $OUT/pytest_kernel.log 2>&1
tail -3 $OUT/pytest_kernel.log
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_parity.py -x -q -k "detection or det or c4 or C4" > $OUT/pytest_det.log 2>&1
tail -3 $OUT/pytest_det.log
B="python bench.py --no-cpu-baseline --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 0 1; do
    TIM_AMD_WGRAD_PAIR=$P timeout 600 $B 2>/dev/null | P=$P python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pair', os.environ['P'], d['ms_per_step'], {k:(d[k].get('graph_replay') or {}).get('ms_per_step') for k in ('c4_train','c3','c2b') if k in d})"
  done
done | tee $OUT/c4_pair_ab.txt
