"""Where a GEMM block spends its cycles (tuning build, make TUNING=1): wait+barrier / DMA issue / MFMA steps / epilogue,
from s_memtime counters of wave 0 of every block (variants 1600 = 128x128 tile, 1614 = 160x128 tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime
dev = "cuda:0"; rt = Runtime("bf16"); g = torch.Generator().manual_seed(3)
for (M, N, K) in ((9920, 3072, 1024), (9920, 1024, 1024), (9920, 2048, 1024), (9920, 1024, 2048), (9920, 1024, 3072)):
    A = torch.randn(M, K, generator=g).to(dev).bfloat16(); B = (torch.randn(N, K, generator=g) / 32).to(dev).bfloat16()
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=dev); bias = torch.zeros(N, device=dev)
    for v in [int(x) for x in os.environ.get("VARS", "1600,1614").split(",")]:
        os.environ["TIMHIP_GEMM_VARIANT"] = str(v)
        cnt = torch.zeros(8, dtype=torch.int64, device=dev)
        for _ in range(3): rt.gemm(L.EPI_STORE_T, A, B, M, N, K, out, N, bias=bias, aux=cnt, ldaux=0)
        cnt.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): rt.gemm(L.EPI_STORE_T, A, B, M, N, K, out, N, bias=bias, aux=cnt, ldaux=0)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        c = cnt.cpu().tolist(); nb = max(c[5], 1); nk = K // 64
        print("M%d N%d K%d v%d: %.1fus %.0fTF | per block: total %.0f cyc = wait %.0f + issue %.0f + mma %.0f + epilogue %.0f (+prologue %.0f) | per k-step: wait %.0f issue %.0f mma %.0f | clock %.2f GHz, block life %.1f us"
              % (M, N, K, v, us, 2.0 * M * N * K / us / 1e6, c[4] / nb, c[0] / nb, c[1] / nb, c[2] / nb, c[3] / nb,
                 (c[4] - c[0] - c[1] - c[2] - c[3]) / nb, c[0] / nb / nk, c[1] / nb / nk, c[2] / nb / nk, c[4] / max(c[6], 1) * 0.1, c[6] / nb / 100.0), flush=True)
