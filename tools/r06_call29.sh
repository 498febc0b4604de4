#!/bin/bash
# round 6, GPU call 29: weight refresh writing the first layers' 16-bit copies last (TIM_AMD_CAST_REVERSE=1), in the step
TAG=${1:-r06ao}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
B="python bench.py --no-cpu-baseline --no-secondary --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 0 1; do
    TIM_AMD_CAST_REVERSE=$P timeout 300 $B 2>/dev/null | P=$P python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cast_reverse', os.environ['P'], d['ms_per_step'], d['roofline']['frac'], d['forward_only']['ms_per_step'])"
  done
done | tee $OUT/cast_reverse_ab.txt
