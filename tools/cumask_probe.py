"""Probe hipExtStreamCreateWithCUMask: time the weight-gradient GEMM and LayerNorm-backward on streams restricted to subsets of the
CUs (experiment for partitioning the two backward streams).  Tuning tool."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd.functional import Runtime
hip = C.CDLL("libamdhip64.so")
dev = "cuda:0"; torch.cuda.set_device(0); rt = Runtime("bf16"); g = torch.Generator().manual_seed(3)
M, E = 9920, 1024
dY = torch.randn(M, 2 * E, generator=g).to(dev).bfloat16(); X = torch.randn(M, E, generator=g).to(dev).bfloat16()
dW = torch.zeros((2 * E, E), device=dev); db = torch.zeros(2 * E, device=dev)
dx = torch.randn(M, E, device=dev); y = torch.randn(M, E, device=dev); stats = torch.rand(M, 2, device=dev) + 0.5
w = torch.ones(E, device=dev); dyf = torch.empty(M, E, device=dev); dyt = torch.empty(M, E, device=dev, dtype=torch.bfloat16)
dg = torch.zeros(E, device=dev); dbt = torch.zeros(E, device=dev)
def masked_stream(bits):
    n = 8   # 256 bits
    arr = (C.c_uint32 * n)(*[(bits >> (32 * i)) & 0xffffffff for i in range(n)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), n, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)
def timeit(stream, f, n=20):
    with torch.cuda.stream(stream):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
wg = lambda: rt.wgrad(dY, 2 * E, X, E, M, dW, db)
ln = lambda: rt.ln_bwd(dx, y, stats, M, E, 0, w, dyf=dyf, dyt=dyt, dgamma=dg, dbeta=dbt)
full = (1 << 256) - 1
masks = {"all 256": full, "low 192": (1 << 192) - 1, "low 128": (1 << 128) - 1, "low 64": (1 << 64) - 1,
         "every 4th off (192)": int("".join("0111" for _ in range(64)), 2), "even (128)": int("01" * 128, 2),
         "every 4th on (64)": int("0001" * 64, 2)}
for name, m in masks.items():
    st = masked_stream(m)
    print("%-22s wgrad %.1f us   ln_bwd %.1f us" % (name, timeit(st, wg), timeit(st, ln)), flush=True)

# ---- NT GEMM on CU subsets, and two GEMMs on disjoint halves at the same time
from tim_amd import _lib as L
A = torch.randn(M, E, generator=g).to(dev).bfloat16(); Bw = (torch.randn(3 * E, E, generator=g) / 32).to(dev).bfloat16()
out = torch.zeros((M, 3 * E), dtype=torch.bfloat16, device=dev); bias = torch.zeros(3 * E, device=dev)
out2 = torch.zeros_like(out)
nt = lambda o=out: rt.gemm(L.EPI_STORE_T, A, Bw, M, 3 * E, E, o, 3 * E, bias=bias)
lo128, hi128 = masked_stream((1 << 128) - 1), masked_stream(((1 << 128) - 1) << 128)
lo192, hi64 = masked_stream((1 << 192) - 1), masked_stream(((1 << 64) - 1) << 192)
allc = masked_stream(full)
for name, st in (("all", allc), ("low 192", lo192), ("low 128", lo128), ("high 128", hi128), ("high 64", hi64)):
    print("NT in_proj fwd on %-9s %.1f us" % (name, timeit(st, nt)), flush=True)
def both(s1, f1, s2, f2, n=20):
    for _ in range(2):
        with torch.cuda.stream(s1): f1()
        with torch.cuda.stream(s2): f2()
    torch.cuda.synchronize()
    e0, e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    s1.wait_event(e0); s2.wait_event(e0)
    with torch.cuda.stream(s1):
        for _ in range(n): f1()
        e1.record()
    with torch.cuda.stream(s2):
        for _ in range(n): f2()
        e2.record()
    torch.cuda.synchronize()
    return max(e0.elapsed_time(e1), e0.elapsed_time(e2)) / n * 1e3
print("two NT GEMMs at once: halves %.1f us per pair ; both on all CUs (two plain streams) %.1f us per pair ; one after the other %.1f us"
      % (both(lo128, nt, hi128, lambda: nt(out2)), both(allc, nt, masked_stream(full), lambda: nt(out2)), 2 * timeit(allc, nt)), flush=True)
print("NT + wgrad at once: halves %.1f us ; all/all %.1f us ; serial %.1f us"
      % (both(lo128, nt, hi128, wg), both(allc, nt, masked_stream(full), wg), timeit(allc, nt) + timeit(allc, wg)), flush=True)
print("ln_bwd (high 64) + wgrad (low 192) at once: %.1f us ; all/all %.1f us ; serial %.1f us"
      % (both(lo192, wg, hi64, ln), both(allc, wg, masked_stream(full), ln), timeit(allc, ln) + timeit(allc, wg)), flush=True)

# ---- one full-batch GEMM vs two half-batch GEMMs on two streams (would a two-half-batch pipeline pay?)
Mh = M // 2
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for (N_, K_, name) in ((3 * E, E, "in_proj fwd"), (E, E, "out_proj dgrad"), (E, 2 * E, "ffn1 dgrad K2048")):
    A_ = torch.randn(M, K_, generator=g).to(dev).bfloat16(); B_ = (torch.randn(N_, K_, generator=g) / 32).to(dev).bfloat16()
    o_ = torch.zeros((M, N_), dtype=torch.bfloat16, device=dev); b_ = torch.zeros(N_, device=dev)
    fullf = lambda: rt.gemm(L.EPI_STORE_T, A_, B_, M, N_, K_, o_, N_, bias=b_)
    h1 = lambda: rt.gemm(L.EPI_STORE_T, A_[:Mh], B_, Mh, N_, K_, o_[:Mh], N_, bias=b_)
    h2 = lambda: rt.gemm(L.EPI_STORE_T, A_[Mh:], B_, Mh, N_, K_, o_[Mh:], N_, bias=b_)
    print("%-18s full %.1f us ; two halves concurrently %.1f us ; two halves back to back %.1f us"
          % (name, timeit(s1, fullf), both(s1, h1, s2, h2), timeit(s1, h1) + timeit(s1, h2)), flush=True)
