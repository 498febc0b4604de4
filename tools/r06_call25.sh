#!/bin/bash
# round 6, GPU call 25: two phases per step in the eight-phase NT kernel (TIMHIP_GEMM_P8_PH=2)
TAG=${1:-r06af}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_eight_phase_kernel" > $OUT/pytest_kernel.log 2>&1
tail -3 $OUT/pytest_kernel.log
timeout 400 python tools/p8_ab.py 5 > $OUT/p8_ab.txt 2>&1
cat $OUT/p8_ab.txt | cut -c1-190
B="python bench.py --no-cpu-baseline --no-secondary --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 4 2; do
    TIMHIP_GEMM_P8_PH=$P timeout 300 $B 2>/dev/null | P=$P python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt_p8_phases', os.environ['P'], d['ms_per_step'], d['roofline']['frac'], d['forward_only']['ms_per_step'])"
  done
done | tee $OUT/step_ab.txt
