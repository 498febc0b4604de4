"""Where the ping-pong GEMM kernel (gemm_pp.hip) spends its cycles - tuning build (make -C tim_amd/csrc TUNING=1), per wave
group: LOAD phase (fragment reads + waits), wait at the barrier that ends it, MFMA phase (20 MFMAs + DMA pieces), wait at
the barrier that ends it; effective shader clock from s_memtime / s_memrealtime."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime
os.environ["TIMHIP_PP_PROF"] = "1"
dev = "cuda:0"; rt = Runtime(os.environ.get("PREC", "fp16")); g = torch.Generator().manual_seed(3)
for (M, N, K) in ((9920, 1024, 3072), (9920, 3072, 1024), (9920, 1024, 1024), (9920, 2048, 1024), (9920, 1024, 2048)):
    A = torch.randn(M, K, generator=g).to(dev).to(rt.op_dtype); B = (torch.randn(N, K, generator=g) / 32).to(dev).to(rt.op_dtype)
    out = torch.zeros((M, N), dtype=rt.op_dtype, device=dev); bias = torch.zeros(N, device=dev)
    cnt = torch.zeros(16, dtype=torch.int64, device=dev)
    for _ in range(3): rt.gemm(L.EPI_STORE_T, A, B, M, N, K, out, N, bias=bias, aux=cnt, ldaux=0)
    cnt.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): rt.gemm(L.EPI_STORE_T, A, B, M, N, K, out, N, bias=bias, aux=cnt, ldaux=0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    c = cnt.cpu().tolist(); nk = K // 64
    print("M%d N%d K%d: %.1f us %.0f TF" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
    for gi, name in ((0, "G0"), (8, "G1")):
        nb = max(c[gi + 5], 1)
        ph = nk   # one LOAD + one MFMA phase per contraction step
        print("   %s per step: load %.0f | barrier %.0f | mfma %.0f | barrier %.0f  (sum %.0f cyc per contraction step and wave; 40 MFMAs = 640 cyc of matrix pipe)   block: "
              "loop+prologue %.0f cyc, epilogue %.0f cyc, clock %.2f GHz, life %.1f us"
              % (name, c[gi] / nb / ph, c[gi + 1] / nb / ph, c[gi + 2] / nb / ph, c[gi + 3] / nb / ph,
                 (c[gi] + c[gi + 1] + c[gi + 2] + c[gi + 3]) / nb / ph, c[gi + 4] / nb, c[gi + 7] / nb,
                 (c[gi + 4] + c[gi + 7]) / max(c[gi + 6], 1) * 0.1, c[gi + 6] / nb / 100.0), flush=True)
