#!/bin/bash
# round 6, GPU call 8: attention backward, first-round blocks touching the next round's first operands (TIMHIP_ATTN_PFN): A/B
TAG=${1:-r06h}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
TIMHIP_ATTN_PFN=1 timeout 600 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_kernels.py -x -q -k "c2a_train_mode or attention or attn" > $OUT/pytest_subset.log 2>&1
tail -2 $OUT/pytest_subset.log
for P in 0 1 0 1 0 1; do
  TIMHIP_ATTN_PFN=$P timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-per-shape --steps 20 --warmup 5 > $OUT/bench_pfn_${P}_$RANDOM.json 2> /dev/null
done
TAG=$TAG python - <<'PY'
import json, os, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/%s/bench_pfn_*.json" % os.environ["TAG"])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["ms_per_step"], d["repeat_ms"], d["non_gemm"]["attention"])
    except Exception as e:
        print(f, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-step --no-per-shape --no-secondary --no-repeat --no-roofline"
for P in 0 1; do
  TIMHIP_ATTN_PFN=$P timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof$P -o c2a -- $B --steps 30 --warmup 5 > /dev/null 2>&1
  python /root/repo/tools/rocpd_stats.py $(find $OUT/prof$P -name "*.db" | head -1) 2>/dev/null | grep -E "attn_|TOTAL" | cut -c1-200 | sed "s/^/PFN=$P /"
  rm -rf $OUT/prof$P
done
