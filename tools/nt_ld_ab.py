"""A/B of the two one-block-per-CU NT GEMM kernels (gemm_pp.hip): 8 waves with the DMA pieces between their MFMAs (TIMHIP_GEMM_LD=0)
against 8 consumer + 4 loader waves (TIMHIP_GEMM_LD=1), interleaved in one process, on the encoder layer's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime
dev = "cuda:0"; rt = Runtime(os.environ.get("PREC", "fp16")); g = torch.Generator().manual_seed(3)
for (M, N, K) in ((9920, 1024, 3072), (9920, 3072, 1024), (9920, 1024, 1024), (9920, 2048, 1024), (9920, 1024, 2048)):
    A = torch.randn(M, K, generator=g).to(dev).to(rt.op_dtype); B = (torch.randn(N, K, generator=g) / 32).to(dev).to(rt.op_dtype)
    oT = torch.zeros((M, N), dtype=rt.op_dtype, device=dev); oF = torch.zeros((M, N), device=dev); bias = torch.zeros(N, device=dev)
    res = torch.randn(M, N, generator=g).to(dev)
    def run(epi):
        if epi == "store_t": rt.gemm(L.EPI_STORE_T, A, B, M, N, K, oT, N, bias=bias)
        else: rt.gemm(L.EPI_ADD_F32, A, B, M, N, K, oF, N, res=res, ldres=N)
    for epi in ("store_t", "add_f32"):
        line = "M%d N%d K%d %s:" % (M, N, K, epi)
        for rep in range(3):
            for ld in ("0", "1"):
                os.environ["TIMHIP_GEMM_LD"] = ld; L.reload_env()
                for _ in range(3): run(epi)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): run(epi)
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 20 * 1e3
                line += "  LD%s %.1f us (%.0f TF)" % (ld, us, 2.0 * M * N * K / us / 1e6)
        print(line, flush=True)
