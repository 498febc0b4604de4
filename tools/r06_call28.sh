#!/bin/bash
# round 6, GPU call 28: linear2's input gradient (multiply-by-saved-factor epilogue) on the two-phase 320 x 256 kernel (TIMHIP_GEMM_P8=2)
# against the tile walk, in the step
TAG=${1:-r06an}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
B="python bench.py --no-cpu-baseline --no-secondary --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 1 2; do
    TIMHIP_GEMM_P8=$P timeout 300 $B 2>/dev/null | P=$P python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gemm_p8', os.environ['P'], d['ms_per_step'], d['roofline']['frac'])"
  done
done | tee $OUT/p8_mulaux_step_ab.txt
