#!/bin/bash
# round 6, GPU call 30: cached instead of streaming stores of the paired weight gradients (experiment)
TAG=${1:-r06aq}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
B="python bench.py --no-cpu-baseline --no-secondary --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  timeout 300 $B 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streaming', d['ms_per_step'], d['roofline']['frac'])"
  TIMHIP_W8_PLAIN_STORES=1 timeout 300 $B 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cached   ', d['ms_per_step'], d['roofline']['frac'])"
done | tee $OUT/w8_store_ab.txt
