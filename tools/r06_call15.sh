#!/bin/bash
# round 6, GPU call 15: eight-phase weight-gradient kernel with the bias MFMAs spread over blocks and waves
TAG=${1:-r06p}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad_group" > $OUT/pytest_kernel.log 2>&1
tail -3 $OUT/pytest_kernel.log
timeout 300 python tools/wg_pair_ab.py 7 > $OUT/wg_pair_ab.txt 2>&1
tail -4 $OUT/wg_pair_ab.txt
B="python bench.py --no-cpu-baseline --no-secondary --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 0 1; do
    TIM_AMD_WGRAD_PAIR=$P timeout 300 $B 2>/dev/null | P=$P python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair', os.environ['P'], d['ms_per_step'], d['roofline']['frac'])"
  done
done | tee $OUT/step_ab.txt
