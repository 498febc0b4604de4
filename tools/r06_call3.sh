#!/bin/bash
# round 6, GPU call 3: full suite with the eight-phase kernel and the attention keep-bits on by default, A/B of the keep-bits in the
# step, kernel stats, the data-parallel wrapper with stubbed collectives
TAG=${1:-r06c}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for KB in 1 0 1 0; do
  TIM_AMD_ATTN_KEEP_BITS=$KB timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-per-shape --steps 20 --warmup 5 > $OUT/bench_kb_${KB}_$RANDOM.json 2> /dev/null
done
TAG=$TAG python - <<'PY'
import json, os, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/%s/bench_kb_*.json" % os.environ["TAG"])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["ms_per_step"], d["repeat_ms"], d["roofline"]["frac"], d.get("non_gemm", {}).get("attention"), d.get("launches_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-step --no-per-shape --no-secondary --no-repeat"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o c2a -- $B --steps 35 --warmup 5 > $OUT/bench_profiled_run.json 2> /dev/null
cd /root/repo
python tools/rocpd_stats.py $(find $OUT/prof -name "*.db" | head -1) > $OUT/kernel_stats.csv 2> $OUT/kernel_stats.err
rm -rf $OUT/prof
head -22 $OUT/kernel_stats.csv | cut -c1-160
timeout 900 python tools/dp_graph_check.py --short > $OUT/dp_single_gpu_overhead.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $OUT/dp_single_gpu_overhead.txt | tail -16
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
