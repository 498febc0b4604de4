#!/bin/bash
# round 6, GPU call 16: LayerNorm backward rows per block 16 (620 blocks = 2.42 per CU) against 20 (496 = 1.94 per CU), in the step
TAG=${1:-r06s}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
B="python bench.py --no-cpu-baseline --no-secondary --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 0 20 0 40; do
    TIMHIP_LN_RPB=$P timeout 300 $B 2>/dev/null | P=$P python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ln_rpb', os.environ['P'], d['ms_per_step'], d['non_gemm']['layernorm']['us_per_step'])"
  done
done | tee $OUT/ln_rpb_step_ab.txt
