"""A/B of the eight-phase NT kernel (gemm_nt_p8_kernel, TIMHIP_GEMM_P8) against the loader-wave kernels on the layer's three
multi-round products, interleaved in ONE process (cdna_hip_programming.md section 5.4 rule 24): per shape and arm the median and
the minimum of R rounds of n back-to-back launches, with a plain 16-bit store and with the epilogue the layer fuses; hipBLASLt
(torch.mm, plain stores) beside them.  Random operands.   python tools/p8_ab.py [rounds] > profiles/r06_p8_ab.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime
dev = "cuda:0"
rt = Runtime("fp16")
M, E, FF = 9920, 1024, 2048
R = int(sys.argv[1]) if len(sys.argv) > 1 else 7
shapes = [("in_proj fwd", 3 * E, E, L.EPI_STORE_T), ("ffn1 fwd", FF, E, L.EPI_GELU_DROP_G2), ("ffn2 dgrad", FF, E, L.EPI_MULAUX_T)]
arms = [("ld/ldp (P8=0)", "0"), ("p8 by shape (P8=1)", "1"), ("p8 256 rows (P8=8)", "8"), ("p8 320 rows (P8=10)", "10"),
        ("p8 256 rows, two phases", "8:2"), ("p8 320 rows, two phases", "10:2")]


def timeit(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def setp8(v):
    os.environ["TIMHIP_GEMM_P8"] = v.split(":")[0]
    os.environ["TIMHIP_GEMM_P8_PH"] = v.split(":")[1] if ":" in v else "4"   # (phases per contraction step)
    L.reload_env()


g = torch.Generator().manual_seed(3)
print("box: %s, %d rounds x 20 launches per arm, interleaved; us = median (min)" % (torch.cuda.get_device_name(0), R))
for name, N, K, epi in shapes:
    A = torch.randn(M, K, generator=g).to(dev).half()
    B = (torch.randn(N, K, generator=g) / 32).to(dev).half()
    oT = torch.zeros((M, N), dtype=torch.float16, device=dev)
    o0 = torch.zeros((M, N), dtype=torch.float16, device=dev)
    o1 = torch.zeros((M, N), dtype=torch.float16, device=dev)
    bias = torch.zeros(N, device=dev)
    bits = torch.full((M, N // 8), 255, dtype=torch.uint8, device=dev)
    aux = torch.randn(M, N, generator=g).to(dev).half()
    kw = dict(bias=None if epi == L.EPI_MULAUX_T else bias)
    if epi == L.EPI_GELU_DROP_G2:
        kw.update(out1=o1, ld1=N, p_drop=0.1, seed=7, site=5, mask=bits, ldmask=N // 8)
    if epi == L.EPI_MULAUX_T:
        kw.update(aux=aux, ldaux=N)
    res = {a: {"plain": [], "fused": []} for a, _ in arms}
    lib = []
    for r in range(R):
        for a, v in arms:
            setp8(v)
            res[a]["plain"].append(timeit(lambda: rt.gemm(L.EPI_STORE_T, A, B, M, N, K, oT, N, bias=bias)))
            res[a]["fused"].append(timeit(lambda: rt.gemm(epi, A, B, M, N, K, o0, N, **kw)))
        lib.append(timeit(lambda: torch.mm(A, B.t())))
    fl = 2.0 * M * N * K
    med = lambda v: sorted(v)[len(v) // 2]
    print("\n%s  %d x %d x %d   hipBLASLt %.1f (%.1f) us = %.0f TF" % (name, M, N, K, med(lib), min(lib), fl / med(lib) / 1e6))
    for a, v in arms:
        setp8(v)
        ch = L.load().timhip_gemm_p8_choice(epi, M, N, K)
        p, f = res[a]["plain"], res[a]["fused"]
        print("  %-22s tile %-3s plain store %6.1f (%6.1f) us = %4.0f TF    layer's epilogue %6.1f (%6.1f) us = %4.0f TF"
              % (a, {0: "160", 8: "256", 10: "320"}[ch], med(p), min(p), fl / med(p) / 1e6, med(f), min(f), fl / med(f) / 1e6), flush=True)
setp8("0")
