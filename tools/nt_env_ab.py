"""A/B of NT-kernel variants chosen by environment switches, interleaved in one process, on the eight NT GEMMs of an encoder
layer with the epilogues the layer uses (C2a, B = 64: M = 9920).
    VARIANTS="name:K=V,K=V;name2:K=V" python tools/nt_env_ab.py      (PLAIN=1: plain 16-bit stores on every shape)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime
dev = "cuda:0"; rt = Runtime(os.environ.get("PREC", "fp16")); g = torch.Generator().manual_seed(3)
M = int(os.environ.get("M", "9920"))
E, FF = 1024, 2048
shapes = [("in_proj fwd", 3 * E, E, L.EPI_STORE_T), ("out_proj fwd", E, E, L.EPI_DROP_RES_F32), ("ffn1 fwd", FF, E, L.EPI_GELU_DROP_G2),
          ("ffn2 fwd", E, FF, L.EPI_DROP_RES_F32), ("ffn2 dgrad", FF, E, L.EPI_MULAUX_T), ("ffn1 dgrad", E, FF, L.EPI_STORE_T),
          ("out_proj dgrad", E, E, L.EPI_STORE_T), ("in_proj dgrad", E, 3 * E, L.EPI_STORE_T)]
default = "pp:TIMHIP_GEMM_LD=0;ld:TIMHIP_GEMM_LD=1,TIMHIP_GEMM_PF=0;ldpf4:TIMHIP_GEMM_LD=1,TIMHIP_GEMM_PF=4;ldpf4_1bar:TIMHIP_GEMM_LD=1,TIMHIP_GEMM_PF=4,TIMHIP_GEMM_LD1=1"
variants = []
for v in os.environ.get("VARIANTS", default).split(";"):
    name, kv = v.split(":")
    variants.append((name, dict(x.split("=") for x in kv.split(","))))
plain = os.environ.get("PLAIN", "0") == "1"
tot = {}
for name, N, K, epi in shapes:
    if plain: epi = L.EPI_STORE_T
    A = torch.randn(M, K, generator=g).to(dev).to(rt.op_dtype); B = (torch.randn(N, K, generator=g) / 32).to(dev).to(rt.op_dtype)
    o0 = torch.zeros((M, N), dtype=torch.float32, device=dev); o1 = torch.zeros((M, N), dtype=torch.float32, device=dev)
    res = torch.randn(M, N, generator=g).to(dev); bias = torch.zeros(N, device=dev)
    stats = torch.ones((M, 2), device=dev); lnw = torch.ones(N, device=dev); lnb = torch.zeros(N, device=dev)
    bits = torch.full((M, N // 8), 255, dtype=torch.uint8, device=dev)
    kw = dict(bias=None if epi == L.EPI_MULAUX_T else bias)
    if epi == L.EPI_DROP_RES_F32: kw.update(res=res, ldres=N, p_drop=0.1, seed=7, site=5, ln=(stats, lnw, lnb))
    if epi == L.EPI_GELU_DROP_G2: kw.update(out1=o1, ld1=N, p_drop=0.1, seed=7, site=5, mask=bits, ldmask=N // 8)
    if epi == L.EPI_MULAUX_T: kw.update(aux=o1, ldaux=N)
    run = lambda: rt.gemm(epi, A, B, M, N, K, o0, N, **kw)
    line = "%-15s N%d K%d:" % (name, N, K)
    best = {}
    for rep in range(4):
        for vname, env in variants:
            os.environ.update(env); L.reload_env()
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            best[vname] = min(best.get(vname, 1e9), e0.elapsed_time(e1) / 20 * 1e3)
    for vname, _ in variants:
        tot[vname] = tot.get(vname, 0.0) + best[vname]
        line += "  %s %.1f (%.0f TF)" % (vname, best[vname], 2.0 * M * N * K / best[vname] / 1e6)
    print(line, flush=True)
print("layer total (us): " + "  ".join("%s %.1f" % kv for kv in tot.items()))
