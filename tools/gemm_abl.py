import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime
dev = "cuda:0"; rt = Runtime("bf16"); g = torch.Generator().manual_seed(3)
for (M, N, K) in ((9920, 3072, 1024), (9920, 1024, 1024), (9920, 2048, 1024), (9920, 1024, 2048), (9920, 1024, 3072)):
    A = torch.randn(M, K, generator=g).to(dev).bfloat16(); B = (torch.randn(N, K, generator=g) / 32).to(dev).bfloat16()
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=dev); bias = torch.zeros(N, device=dev)
    row = []
    for v in [int(x) for x in os.environ.get("VARS","0,100,200,300,700").split(",")]:
        os.environ["TIMHIP_GEMM_VARIANT"] = str(v)
        for _ in range(3): rt.gemm(L.EPI_STORE_T, A, B, M, N, K, out, N, bias=bias)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): rt.gemm(L.EPI_STORE_T, A, B, M, N, K, out, N, bias=bias)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        row.append("abl%d %.1fus %.0fTF" % (v, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
    print(M, N, K, " | ".join(row), flush=True)
