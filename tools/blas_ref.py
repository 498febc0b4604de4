"""Calibration: the ROCm library GEMM (torch.mm -> hipBLASLt/rocBLAS, bf16 in, fp32 accumulate, NO fused epilogue) on the
encoder's shapes, to compare with the hand-written kernels' per-shape numbers in bench.py.  Not used by the product."""
import torch
dev = "cuda:0"
M, E, FF = 9920, 1024, 2048
nt = [("in_proj fwd", M, 3 * E, E), ("out_proj fwd", M, E, E), ("ffn1 fwd", M, FF, E), ("ffn2 fwd", M, E, FF),
      ("in_proj dgrad", M, E, 3 * E)]
tn = [("in_proj wgrad", 3 * E, E, M), ("out_proj wgrad", E, E, M), ("ffn1 wgrad", FF, E, M), ("ffn2 wgrad", E, FF, M)]
def timeit(f):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30
for name, m, n, k in nt:
    A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16()
    ms = timeit(lambda: torch.mm(A, B.t()))
    print("%-16s NT M%d N%d K%d  %.1f us  %.0f TF" % (name, m, n, k, ms * 1e3, 2.0 * m * n * k / ms / 1e9), flush=True)
for name, m, n, k in tn:
    Y = torch.randn(k, m, device=dev).bfloat16(); X = torch.randn(k, n, device=dev).bfloat16()
    ms = timeit(lambda: torch.mm(Y.t(), X))
    print("%-16s TN M%d N%d K%d  %.1f us  %.0f TF" % (name, m, n, k, ms * 1e3, 2.0 * m * n * k / ms / 1e9), flush=True)
