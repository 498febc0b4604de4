#!/bin/bash
# round 6, GPU call 18: LayerNorm forward rows per block (4 = 2480 blocks, one row per wave) against 8 / 12 / 20, in the step
TAG=${1:-r06u}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
B="python bench.py --no-cpu-baseline --no-secondary --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 0 8 0 12 0 20; do
    TIMHIP_LN_FWD_RPB=$P timeout 300 $B 2>/dev/null | P=$P python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ln_fwd_rpb', os.environ['P'], d['ms_per_step'], d['non_gemm']['layernorm']['us_per_step'], d['forward_only']['ms_per_step'])"
  done
done | tee $OUT/ln_fwd_rpb_step_ab.txt
