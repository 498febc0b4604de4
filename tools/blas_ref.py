"""Calibration: the ROCm library GEMM (torch.mm -> hipBLASLt / rocBLAS, 16-bit in, fp32 accumulate, plain stores, NO fused
epilogue) against the hand-written kernels on the encoder layer's shapes, same process, same box, same dtype (fp16).
NT: gemm_nt_pp_kernel with a plain 16-bit store (TIMHIP_EPI_STORE_T) and with the epilogue the layer actually fuses;
TN (weight gradients): the four library GEMMs one by one against the layer's ONE grouped launch (wgrad_ld_kernel).
Not used by the product.   python tools/blas_ref.py > profiles/rNN_blas_ref.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime
dev = "cuda:0"
rt = Runtime("fp16")
M, E, FF = 9920, 1024, 2048
nt = [("in_proj fwd", 3 * E, E, L.EPI_STORE_T), ("out_proj fwd", E, E, L.EPI_DROP_RES_F32), ("ffn1 fwd", FF, E, L.EPI_GELU_DROP_G2),
      ("ffn2 fwd", E, FF, L.EPI_DROP_RES_F32), ("ffn2 dgrad", FF, E, L.EPI_MULAUX_T), ("ffn1 dgrad", E, FF, L.EPI_STORE_T),
      ("out_proj dgrad", E, E, L.EPI_STORE_T), ("in_proj dgrad", E, 3 * E, L.EPI_STORE_T)]
tn = [("ffn2 wgrad", E, FF), ("ffn1 wgrad", FF, E), ("out_proj wgrad", E, E), ("in_proj wgrad", 3 * E, E)]
def timeit(f, n=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(3)
print("%-16s %-18s %14s %20s %26s" % ("NT GEMM", "M x N x K", "hipBLASLt", "hand, plain store", "hand, the layer's epilogue"))
tl = th = tf = 0.0
for name, N, K, epi in nt:
    A = torch.randn(M, K, generator=g).to(dev).half(); B = (torch.randn(N, K, generator=g) / 32).to(dev).half()
    us_lib = timeit(lambda: torch.mm(A, B.t()))
    oT = torch.zeros((M, N), dtype=torch.float16, device=dev); o0 = torch.zeros((M, N), device=dev); o1 = torch.zeros((M, N), device=dev)
    bias = torch.zeros(N, device=dev); res = torch.randn(M, N, generator=g).to(dev)
    stats = torch.ones((M, 2), device=dev); lnw = torch.ones(N, device=dev); lnb = torch.zeros(N, device=dev)
    bits = torch.full((M, N // 8), 255, dtype=torch.uint8, device=dev)
    us_plain = timeit(lambda: rt.gemm(L.EPI_STORE_T, A, B, M, N, K, oT, N, bias=bias))
    kw = dict(bias=None if epi == L.EPI_MULAUX_T else bias)
    if epi == L.EPI_DROP_RES_F32: kw.update(res=res, ldres=N, p_drop=0.1, seed=7, site=5, ln=(stats, lnw, lnb))
    if epi == L.EPI_GELU_DROP_G2: kw.update(out1=o1, ld1=N, p_drop=0.1, seed=7, site=5, mask=bits, ldmask=N // 8)
    if epi == L.EPI_MULAUX_T: kw.update(aux=o1, ldaux=N)
    us_fused = timeit(lambda: rt.gemm(epi, A, B, M, N, K, o0, N, **kw))
    fl = 2.0 * M * N * K
    tl += us_lib; th += us_plain; tf += us_fused
    print("%-16s %-18s %7.1f us %4.0f TF %12.1f us %4.0f TF %18.1f us %4.0f TF" % (name, "%dx%dx%d" % (M, N, K), us_lib, fl / us_lib / 1e6,
          us_plain, fl / us_plain / 1e6, us_fused, fl / us_fused / 1e6), flush=True)
print("%-35s %7.1f us %20.1f us %26.1f us" % ("sum of the eight", tl, th, tf))
print()
items, fl, t_lib = [], 0.0, 0.0
for name, no, ko in tn:
    Y = torch.randn(M, no, generator=g).to(dev).half(); X = torch.randn(M, ko, generator=g).to(dev).half()
    us = timeit(lambda: torch.mm(Y.t(), X))
    t_lib += us
    print("%-16s TN %dx%dx%d  hipBLASLt %.1f us  %.0f TF" % (name, no, ko, M, us, 2.0 * no * ko * M / us / 1e6), flush=True)
    items.append((Y, no, X, ko, torch.zeros((no, ko), device=dev), torch.zeros(no, device=dev)))
    fl += 2.0 * no * ko * M
us = timeit(lambda: rt.wgrad_group(items, M, accumulate=False), 20)
print("the four as the layer's ONE grouped launch (wgrad_ld_kernel, bias gradients included): %.1f us  %.0f TF   (hipBLASLt, four launches: %.1f us  %.0f TF)"
      % (us, fl / us / 1e6, t_lib, fl / t_lib / 1e6))
items2 = items + [(Y, no, X, ko, torch.zeros((no, ko), device=dev), torch.zeros(no, device=dev)) for (Y, no, X, ko, _, _) in items]
us2 = timeit(lambda: rt.wgrad_group(items2, M, accumulate=False), 20)
print("round 6: TWO layers' eight products as one round of 256 eight-phase 256 x 256 tiles (wgrad_p8_kernel): %.1f us = %.1f us per layer  %.0f TF"
      % (us2, us2 / 2, 2 * fl / us2 / 1e6))
