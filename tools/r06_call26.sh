#!/bin/bash
# round 6, GPU call 26: waves per attention-forward block (4 = default: 5 row blocks on 4 waves; 5; 8) in the step
TAG=${1:-r06ai}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
B="python bench.py --no-cpu-baseline --no-secondary --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2; do
  for P in 0 5 0 8 0 3; do
    TIMHIP_ATTN_WAVES=$P timeout 300 $B 2>/dev/null | P=$P python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('attn_waves', os.environ['P'], d['ms_per_step'], d['non_gemm']['attention']['us_per_step'], d['forward_only']['ms_per_step'])"
  done
done | tee $OUT/attn_waves_ab.txt
