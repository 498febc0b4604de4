#!/bin/bash
# round 6, GPU call 11: the dropout + residual epilogue's keep-bits drawn inside the main loop (TIMHIP_EPI_PAIR=2) against the paired
# draws in the epilogue (1): tests, isolated and in-step A/B
TAG=${1:-r06k}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
TIMHIP_EPI_PAIR=2 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_parity.py -x -q -k "pingpong_kernel or train_mode_vs_oracle or operating_points" > $OUT/pytest_subset.log 2>&1
tail -2 $OUT/pytest_subset.log
VARIANTS="loop:TIMHIP_EPI_PAIR=2;pair:TIMHIP_EPI_PAIR=1;own:TIMHIP_EPI_PAIR=0" timeout 600 python tools/nt_env_ab.py 2>&1 | grep -E "out_proj fwd|ffn2 fwd|layer total"
for P in 2 1 2 1 2 1; do
  TIMHIP_EPI_PAIR=$P timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-per-shape --steps 20 --warmup 5 > $OUT/bench_epi_${P}_$RANDOM.json 2> /dev/null
done
TAG=$TAG python - <<'PY'
import json, os, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/%s/bench_epi_*.json" % os.environ["TAG"])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["ms_per_step"], d["repeat_ms"], d["roofline"]["frac"], d["forward_only"]["ms_per_step"])
    except Exception as e:
        print(f, "failed", e)
PY
