#!/bin/bash
# round 6, GPU call 4: keep-bits launch on the side stream (A/B), knob sweep of the c2a_b8 operating point (8 windows per GPU)
TAG=${1:-r06d}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_graph.py tests/test_gpu_kernels.py -x -q -k "keep_bits or layernorm or graph or c2a_train_mode or operating_points or replay" > $OUT/pytest_subset.log 2>&1
tail -3 $OUT/pytest_subset.log
for SIDE in 0 1 0 1; do
  TIM_AMD_KEEP_BITS_SIDE=$SIDE timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-per-shape --steps 20 --warmup 5 > $OUT/bench_side_${SIDE}_$RANDOM.json 2> /dev/null
done
b8() {  # label, env...
  local label=$1; shift
  local line=$(env "$@" timeout 200 python bench.py --graph-child --workload C2a --batch 8 --steps 40 --warmup 10 2>/dev/null | tail -1)
  echo "$label :: $line"
}
{
  echo "c2a_b8 (C2a, 8 windows per GPU, fp16, HIP-graph replay of the fixed-cotangent forward + backward step): ms per step by knob"
  b8 "default" X=1
  b8 "default (repeat)" X=1
  b8 "TIMHIP_LN_RPB=16 (round-5 block height of LayerNorm backward)" TIMHIP_LN_RPB=16
  b8 "TIMHIP_ATTN_SPLIT_MIN=1" TIMHIP_ATTN_SPLIT_MIN=1
  b8 "TIMHIP_ATTN_SPLIT_MIN=2" TIMHIP_ATTN_SPLIT_MIN=2
  b8 "TIMHIP_GEMM_PP_MIN_TILES=32" TIMHIP_GEMM_PP_MIN_TILES=32
  b8 "TIMHIP_GEMM_PP_MIN_TILES=64" TIMHIP_GEMM_PP_MIN_TILES=64
  b8 "TIMHIP_GEMM_PP_MIN_TILES=96" TIMHIP_GEMM_PP_MIN_TILES=96
  b8 "TIMHIP_ATTN_FUSED=0 (two-kernel attention backward)" TIMHIP_ATTN_FUSED=0
  b8 "TIMHIP_ATTN_SPLIT_MIN=1 TIMHIP_GEMM_PP_MIN_TILES=32" TIMHIP_ATTN_SPLIT_MIN=1 TIMHIP_GEMM_PP_MIN_TILES=32
} > $OUT/c2a_b8_knobs.txt 2>&1
cat $OUT/c2a_b8_knobs.txt
TAG=$TAG python - <<'PY'
import json, os, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/%s/bench_side_*.json" % os.environ["TAG"])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["ms_per_step"], d["repeat_ms"], d["eager"]["ms_per_step"], d["forward_only"])
    except Exception as e:
        print(f, "failed", e)
PY
