"""What the NT kernels reach at a weight-gradient-sized contraction (K = 9920) on exactly 256 tiles, beside the layer's grouped
weight-gradient launch (wgrad_ld_kernel, 128 x 256 tiles, transposing reads): an upper bound for an eight-phase TN kernel with
256 x 256 tiles.  Interleaved, median (min) of R rounds.   python tools/p8_longk.py [rounds] > profiles/r06_n_p8_longk.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime
dev = "cuda:0"
rt = Runtime("fp16")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
K = 9920 // 64 * 64
E, FF, Mw = 1024, 2048, 9920


def timeit(f, n=10):
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
    os.environ["TIMHIP_GEMM_P8"] = v
    L.reload_env()


g = torch.Generator().manual_seed(5)
arms = [("ld 160 x 256", "0", 2560, 4096), ("p8 256 x 256", "8", 4096, 4096), ("p8 320 x 256", "10", 5120, 4096)]
ops = {}
for name, v, M, N in arms:
    A = torch.randn(M, K, generator=g).to(dev).half()
    B = (torch.randn(N, K, generator=g) / 32).to(dev).half()
    ops[name] = (A, B, torch.zeros((M, N), dtype=torch.float16, device=dev), torch.zeros(N, device=dev))
items, flw = [], 0.0
for no, ko in [(E, FF), (FF, E), (E, E), (3 * E, E)]:
    Y = torch.randn(Mw, no, generator=g).to(dev).half()
    X = torch.randn(Mw, ko, generator=g).to(dev).half()
    items.append((Y, no, X, ko, torch.zeros((no, ko), device=dev), torch.zeros(no, device=dev)))
    flw += 2.0 * no * ko * Mw
res = {a[0]: [] for a in arms}
res["wgrad"] = []
for r in range(R):
    for name, v, M, N in arms:
        setp8(v)
        A, B, o, bias = ops[name]
        res[name].append(timeit(lambda: rt.gemm(L.EPI_STORE_T, A, B, M, N, K, o, N, bias=bias)))
    res["wgrad"].append(timeit(lambda: rt.wgrad_group(items, Mw, accumulate=False)))
setp8("1")
med = lambda v: sorted(v)[len(v) // 2]
print("box: %s; contraction %d, 256 tiles per launch, %d rounds x 10 launches, us = median (min)" % (torch.cuda.get_device_name(0), K, R))
for name, v, M, N in arms:
    fl = 2.0 * M * N * K
    t = res[name]
    print("  NT %-14s %5d x %5d  %7.1f (%7.1f) us = %5.0f TF" % (name, M, N, med(t), min(t), fl / med(t) / 1e6))
t = res["wgrad"]
print("  TN wgrad_ld 128 x 256, the layer's four gradients in one launch  %7.1f (%7.1f) us = %5.0f TF" % (med(t), min(t), flw / med(t) / 1e6))
