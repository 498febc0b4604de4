#!/bin/bash
# round 6, GPU call 2: the eight-phase NT kernel (tests + interleaved A/B), kernel stats of the c2a_b8 and c2a_train blocks
TAG=${1:-r06b}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "eight_phase or pingpong_row_tile" > $OUT/pytest_p8.log 2>&1
tail -3 $OUT/pytest_p8.log
timeout 600 python tools/p8_ab.py 5 > $OUT/p8_ab.txt 2>&1
cat $OUT/p8_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_b8 -o b8 -- python /root/repo/tools/prof_secondary.py C2a 8 40 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_tr -o tr -- python /root/repo/tools/prof_secondary.py C2a 64 40 --rec-train > /dev/null 2>&1
cd /root/repo
python tools/rocpd_stats.py $(find $OUT/prof_b8 -name "*.db" | head -1) > $OUT/c2a_b8_kernel_stats.csv 2> $OUT/b8.err
python tools/rocpd_stats.py $(find $OUT/prof_tr -name "*.db" | head -1) > $OUT/c2a_train_kernel_stats.csv 2> $OUT/tr.err
python tools/rocpd_timeline.py $(find $OUT/prof_b8 -name "*.db" | head -1) 0 -2 > $OUT/timeline_b8.txt 2>&1
rm -rf $OUT/prof_b8 $OUT/prof_tr
for P8 in 0 1; do
  TIMHIP_GEMM_P8=$P8 timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $OUT/bench_p8_$P8.json 2> $OUT/bench_p8_$P8.err
done
TAG=$TAG python - <<'PY'
import json, os
for p8 in (0, 1):
    try:
        d = json.loads(open("/root/repo/gpurun_out/%s/bench_p8_%d.json" % (os.environ["TAG"], p8)).read().strip().splitlines()[-1])
        print("P8=%d" % p8, d["ms_per_step"], d["repeat_ms"], d["roofline"]["frac"], [(s["gemm"][:12], s["us"]) for s in d["roofline"]["per_shape_isolated"]])
    except Exception as e:
        print("P8=%d" % p8, "failed", e)
PY
head -40 $OUT/c2a_b8_kernel_stats.csv
