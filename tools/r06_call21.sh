#!/bin/bash
# round 6, GPU call 21: the in-projection forward as two one-round launches (TIMHIP_GEMM_SPLIT_N): kernel test, isolated timing, the step
TAG=${1:-r06y}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "column_split or eight_phase_choice" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
python - > $OUT/split_isolated.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, "/root/repo")
from tim_amd import _lib as L
from tim_amd.functional import Runtime
rt = Runtime("fp16"); dev = "cuda:0"
M, N, K = 9920, 3072, 1024
g = torch.Generator().manual_seed(3)
A = torch.randn(M, K, generator=g).to(dev).half(); B = (torch.randn(N, K, generator=g) / 32).to(dev).half()
o = torch.zeros((M, N), dtype=torch.float16, device=dev); bias = torch.zeros(N, device=dev)
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = {"0": [], "1": []}
for r in range(7):
    for v in ("0", "1"):
        os.environ["TIMHIP_GEMM_SPLIT_N"] = v; L.reload_env()
        res[v].append(timeit(lambda: rt.gemm(L.EPI_STORE_T, A, B, M, N, K, o, N, bias=bias)))
med = lambda v: sorted(v)[len(v) // 2]
print("in-projection forward 9920 x 3072 x 1024, 16-bit store + bias, 7 rounds x 20 interleaved: one launch of 468 tiles of 256 x 256: %.1f (%.1f) us; two launches (2048 columns on 320 x 256, 1024 on 160 x 256): %.1f (%.1f) us" % (med(res["0"]), min(res["0"]), med(res["1"]), min(res["1"])))
PY
cat $OUT/split_isolated.txt | tail -2
B="python bench.py --no-cpu-baseline --no-secondary --no-per-shape --no-repeat --steps 20 --warmup 5"
for i in 1 2 3; do
  for P in 0 1; do
    TIMHIP_GEMM_SPLIT_N=$P timeout 300 $B 2>/dev/null | P=$P python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split_n', os.environ['P'], d['ms_per_step'], d['roofline']['frac'], d['forward_only']['ms_per_step'])"
  done
done | tee $OUT/split_step_ab.txt
