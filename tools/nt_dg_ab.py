"""A/B of the one-tile-per-block ping-pong NT kernel (TIMHIP_GEMM_DG=0, TIMHIP_GEMM_PT=0), its persistent-tile form (TIMHIP_GEMM_PT=1) and the dual-group persistent kernel
(TIMHIP_GEMM_DG=1, several group offsets), interleaved in one process, on the eight NT GEMMs of an encoder layer with the
epilogues the layer uses (C2a, B = 64: M = 9920)."""
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
offsets = [int(x) for x in os.environ.get("OFFSETS", "1,5,9,13").split(",")]
tot = {}
for name, N, K, epi in shapes:
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
    for cfgname, env in [("pp", {"TIMHIP_GEMM_DG": "0", "TIMHIP_GEMM_PT": "0"}), ("pt", {"TIMHIP_GEMM_DG": "0", "TIMHIP_GEMM_PT": "1"})] + \
            [("dg%d" % o, {"TIMHIP_GEMM_DG": "1", "TIMHIP_GEMM_DG_OFFSET": str(o), "TIMHIP_GEMM_PT": "0"}) for o in offsets]:
        best = 1e9
        for rep in range(3):
            os.environ.update(env); L.reload_env()
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        tot[cfgname] = tot.get(cfgname, 0.0) + best
        line += "  %s %.1f us (%.0f TF)" % (cfgname, best, 2.0 * M * N * K / best / 1e6)
    print(line, flush=True)
print("layer total: " + "  ".join("%s %.1f us" % kv for kv in tot.items()))
