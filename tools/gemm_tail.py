"""Cost of the partial last round: time of the N=1024 GEMM shapes as a function of M (tile quantisation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime
dev = "cuda:0"; rt = Runtime("bf16"); g = torch.Generator().manual_seed(3)
for (N, K, epi, name) in ((1024, 1024, L.EPI_STORE_T, "store_T"), (1024, 1024, L.EPI_DROP_RES_F32, "drop_res"),
                          (1024, 2048, L.EPI_DROP_RES_F32, "drop_res K2048"), (1024, 3072, L.EPI_ADD_F32, "add K3072"),
                          (2048, 1024, L.EPI_STORE_T, "N2048"), (3072, 1024, L.EPI_STORE_T, "N3072")):
    for scratch in (None,):
        row = []
        for M in (8192, 9920, 12288):
            A = torch.randn(M, K, generator=g).to(dev).bfloat16(); B = (torch.randn(N, K, generator=g) / 32).to(dev).bfloat16()
            out = torch.zeros((M, N), dtype=torch.float32, device=dev); res = torch.zeros((M, N), device=dev); bias = torch.zeros(N, device=dev)
            kw = dict(bias=bias)
            if epi == L.EPI_DROP_RES_F32: kw = dict(bias=bias, res=res, ldres=N, p_drop=0.1, seed=1, site=2)
            if epi == L.EPI_ADD_F32: kw = dict(res=res, ldres=N)
            for _ in range(3): rt.gemm(epi, A, B, M, N, K, out, N, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): rt.gemm(epi, A, B, M, N, K, out, N, **kw)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            tiles = ((M + 127) // 128) * (N // 128)
            row.append("M%d: %d tiles %.1fus %.0fTF" % (M, tiles, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
        print(name, " | ".join(row), flush=True)
