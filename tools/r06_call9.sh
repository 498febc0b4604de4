#!/bin/bash
# round 6, GPU call 9: LayerNorm backward, lane pairs sharing the Philox calls of the dropout keep factors (TIMHIP_LN_PAIR): tests + A/B
TAG=${1:-r06i}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -k "layer or train_mode or operating_points or layernorm or keep_bits" > $OUT/pytest_subset.log 2>&1
tail -2 $OUT/pytest_subset.log
for P in 1 0 1 0 1 0; do
  TIMHIP_LN_PAIR=$P timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-per-shape --steps 20 --warmup 5 > $OUT/bench_lnpair_${P}_$RANDOM.json 2> /dev/null
done
TAG=$TAG python - <<'PY'
import json, os, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/%s/bench_lnpair_*.json" % os.environ["TAG"])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["ms_per_step"], d["repeat_ms"], d["non_gemm"]["layernorm"])
    except Exception as e:
        print(f, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-step --no-per-shape --no-secondary --no-repeat --no-roofline"
for P in 0 1; do
  TIMHIP_LN_PAIR=$P timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof$P -o c2a -- $B --steps 30 --warmup 5 > /dev/null 2>&1
  python /root/repo/tools/rocpd_stats.py $(find $OUT/prof$P -name "*.db" | head -1) 2>/dev/null | grep -E "ln_bwd|ln_fwd|TOTAL" | cut -c1-170 | sed "s/^/LN_PAIR=$P /"
  rm -rf $OUT/prof$P
done
