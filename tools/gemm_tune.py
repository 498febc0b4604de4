"""GEMM tile-variant sweep on the encoder's shapes (tuning tool, not part of the product path)."""
import os, sys, itertools, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime

dev = "cuda:0"
rt = Runtime("bf16")
M, E, FF = 9920, 1024, 2048
Mp = 9984
shapes = [("in_proj fwd", M, 3 * E, E, L.EPI_STORE_T, 1), ("out_proj fwd", M, E, E, L.EPI_DROP_RES_F32, 1),
          ("ffn1-like", M, FF, E, L.EPI_STORE_T, 1), ("ffn2 fwd", M, E, FF, L.EPI_DROP_RES_F32, 1),
          ("in_proj dgrad", M, E, 3 * E, L.EPI_ADD_F32, 1), ("out_proj dgrad", M, E, E, L.EPI_STORE_T, 1)]
g = torch.Generator().manual_seed(3)
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5,6,7,8".split(","))]
for name, m_, n_, k_, epi, sk in shapes:
    A = torch.randn(m_, k_, generator=g).to(dev).bfloat16()
    Bm = (torch.randn(n_, k_, generator=g) * k_ ** -0.5).to(dev).bfloat16()
    o0 = torch.zeros((sk * m_, n_), dtype=torch.float32, device=dev)
    o1 = torch.zeros((m_, n_), dtype=torch.float32, device=dev)
    res = torch.zeros((m_, n_), dtype=torch.float32, device=dev)
    bias = torch.zeros(n_, device=dev)
    row = []
    ref = None
    for v in variants:
        os.environ["TIMHIP_GEMM_VARIANT"] = str(v)
        kw = dict(out1=o1, ld1=n_, bias=None if epi in (L.EPI_ADD_F32, L.EPI_DGELU_T) else bias, res=res, ldres=n_,
                  aux=o1, ldaux=n_, p_drop=0.1, seed=7, site=5, splitk=sk)
        if epi == L.EPI_STORE_F32:
            kw = dict(splitk=sk)
        for _ in range(3):
            rt.gemm(epi, A, Bm, m_, n_, k_, o0, n_, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            rt.gemm(epi, A, Bm, m_, n_, k_, o0, n_, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        got = o0.view(torch.bfloat16).flatten()[:m_ * n_].float() if epi in (L.EPI_STORE_T,) else o0[:m_].clone()
        if ref is None:
            ref = got.clone()
        bad = (got - ref).abs().max().item()
        row.append("v%d %5.1fus %4.0fTF%s" % (v, ms * 1e3, 2.0 * m_ * n_ * k_ / ms / 1e9, "" if bad < 1e-2 else " BAD%.2g" % bad))
    print("%-16s M%5d N%5d K%5d sk%d | " % (name, m_, n_, k_, sk) + " | ".join(row), flush=True)
