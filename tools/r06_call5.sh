#!/bin/bash
# round 6, GPU call 5: residual prefetch of the "+ residual" NT epilogues (TIMHIP_GEMM_RESPF): isolated A/B on the layer's eight
# products and in the step
TAG=${1:-r06e}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "pingpong" > $OUT/pytest_subset.log 2>&1
TIMHIP_GEMM_RESPF=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "pingpong_kernel and True" >> $OUT/pytest_subset.log 2>&1
grep -E "passed|failed" $OUT/pytest_subset.log
VARIANTS="off:TIMHIP_GEMM_RESPF=0;s1:TIMHIP_GEMM_RESPF=1;s2:TIMHIP_GEMM_RESPF=2;s5:TIMHIP_GEMM_RESPF=5;s10:TIMHIP_GEMM_RESPF=10" timeout 600 python tools/nt_env_ab.py > $OUT/respf_isolated_ab.txt 2>&1
cat $OUT/respf_isolated_ab.txt
for RP in 0 2 0 2 5 10; do
  TIMHIP_GEMM_RESPF=$RP timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-per-shape --steps 20 --warmup 5 > $OUT/bench_respf_${RP}_$RANDOM.json 2> /dev/null
done
TAG=$TAG python - <<'PY'
import json, os, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/%s/bench_respf_*.json" % os.environ["TAG"])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["ms_per_step"], d["repeat_ms"], d["roofline"]["frac"], d["forward_only"]["ms_per_step"])
    except Exception as e:
        print(f, "failed", e)
PY
