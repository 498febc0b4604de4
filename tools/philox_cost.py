import os, sys
sys.path.insert(0, "/root/repo")
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime
dev = "cuda:0"; rt = Runtime("bf16"); g = torch.Generator().manual_seed(3)
M = 9920
for (N, K, epi, name) in ((1024, 1024, L.EPI_DROP_RES_F32, "drop_res"), (2048, 1024, L.EPI_GELU_DROP_T2, "gelu_drop"), (2048, 1024, L.EPI_DGELU_T, "dgelu")):
    A = torch.randn(M, K, generator=g).to(dev).bfloat16(); B = (torch.randn(N, K, generator=g) / 32).to(dev).bfloat16()
    out = torch.zeros((M, N), dtype=torch.float32, device=dev); o1 = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
    res = torch.zeros((M, N), device=dev); bias = torch.zeros(N, device=dev)
    mk = torch.randint(0, 256, (M, N // 8), dtype=torch.uint8, device=dev)   # precomputed keep-bits (any bits do for timing)
    for p, use_mask in ((0.0, False), (0.1, False), (0.1, True)):
        kw = dict(bias=bias, res=res, ldres=N, out1=o1, ld1=N, aux=o1, ldaux=N, p_drop=p, seed=1, site=2)
        if use_mask:
            if epi == L.EPI_DROP_RES_F32: continue
            kw.update(mask=mk, ldmask=N // 8)
        if epi == L.EPI_DGELU_T: kw.pop("bias")
        for _ in range(3): rt.gemm(epi, A, B, M, N, K, out, N, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): rt.gemm(epi, A, B, M, N, K, out, N, **kw)
        e1.record(); torch.cuda.synchronize()
        print(name, "p=%.1f%s" % (p, " keep-bits" if use_mask else ""), "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3), flush=True)
