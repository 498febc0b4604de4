#!/bin/bash
# round 6, GPU call 24: two 32-MFMA phases per step in the eight-phase weight-gradient kernel (TIMHIP_WGRAD_P8_PH=2)
TAG=${1:-r06ae}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad_group_eight_phase" > $OUT/pytest_kernel.log 2>&1
tail -3 $OUT/pytest_kernel.log
timeout 300 python tools/wg_pair_ab.py 7 > $OUT/wg_pair_ab.txt 2>&1
tail -5 $OUT/wg_pair_ab.txt
B="python bench.py --no-cpu-baseline --no-secondary --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 4 2; do
    TIMHIP_WGRAD_P8_PH=$P timeout 300 $B 2>/dev/null | P=$P python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p8_phases', os.environ['P'], d['ms_per_step'], d['roofline']['frac'])"
  done
done | tee $OUT/step_ab.txt
