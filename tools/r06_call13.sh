#!/bin/bash
# round 6, GPU call 13: unrolled keep-bits kernel (test + kernel time in the step)
TAG=${1:-r06n}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_train_parity.py -x -q -k "keep_bits or train_mode_vs_oracle" > $OUT/pytest_subset.log 2>&1
tail -2 $OUT/pytest_subset.log
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-step --no-per-shape --no-secondary --no-repeat --no-roofline"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o c2a -- $B --steps 30 --warmup 5 > $OUT/bench_profiled.json 2> /dev/null
python /root/repo/tools/rocpd_stats.py $(find $OUT/prof -name "*.db" | head -1) 2>/dev/null | grep -E "attn_|ln_|TOTAL" | cut -c1-170
rm -rf $OUT/prof
cd /root/repo
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-per-shape --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeat_ms'], d['roofline']['frac'])"; done
