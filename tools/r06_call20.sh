#!/bin/bash
# round 6, GPU call 20: 128 x 128 weight-gradient tiles with a ring of four 32-row half-stages (TIMHIP_WGRAD_RING): kernel tests,
# then the steps against the two 64-row stages
TAG=${1:-r06x}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad" > $OUT/pytest_wgrad.log 2>&1
tail -3 $OUT/pytest_wgrad.log
B="python bench.py --no-cpu-baseline --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 0 1; do
    TIMHIP_WGRAD_RING=$P timeout 600 $B 2>/dev/null | P=$P python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wgrad_ring', os.environ['P'], d['ms_per_step'], d['roofline'].get('traffic'), {k:(d[k].get('graph_replay') or {}).get('ms_per_step') for k in ('c2a_b8','c2a_train','c2b','c1','c3','c4_train') if k in d})"
  done
done | tee $OUT/wgrad_ring_ab.txt
