#!/bin/bash
# round 6, GPU call 17: LayerNorm backward's balanced block height as the default: parity subset, then the step against 16 rows
TAG=${1:-r06t}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_graph.py tests/test_gpu_kernels.py -x -q -k "not gemm" > $OUT/pytest_subset.log 2>&1
tail -3 $OUT/pytest_subset.log
B="python bench.py --no-cpu-baseline --no-secondary --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 16 0; do
    TIMHIP_LN_RPB=$P timeout 300 $B 2>/dev/null | P=$P python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ln_rpb', os.environ['P'], d['ms_per_step'], d['non_gemm']['layernorm']['us_per_step'])"
  done
done | tee $OUT/ln_rpb_step_ab.txt
