#!/bin/bash
# round 6, GPU call 14: eight-phase weight-gradient kernel (two layers per launch): kernel test, train parity, step A/B
TAG=${1:-r06o}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad_group_eight_phase" > $OUT/pytest_kernel.log 2>&1
tail -3 $OUT/pytest_kernel.log
timeout 900 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_graph.py -x -q > $OUT/pytest_subset.log 2>&1
tail -3 $OUT/pytest_subset.log
timeout 300 python tools/wg_pair_ab.py 5 > $OUT/wg_pair_ab.txt 2>&1
cat $OUT/wg_pair_ab.txt | tail -8
B="python bench.py --no-cpu-baseline --no-secondary --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 0 1; do
    TIM_AMD_WGRAD_PAIR=$P timeout 300 $B 2>/dev/null | P=$P python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair', os.environ['P'], d['ms_per_step'], d['roofline']['frac'], d.get('parity'))"
  done
done | tee $OUT/step_ab.txt
