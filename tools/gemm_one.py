"""Run one GEMM shape repeatedly (for rocprofv3 PMC collection).  usage: gemm_one.py nt|tn [variant]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime
kind = sys.argv[1] if len(sys.argv) > 1 else "nt"
if len(sys.argv) > 2: os.environ["TIMHIP_GEMM_VARIANT"] = sys.argv[2]
dev = "cuda:0"; rt = Runtime("bf16"); g = torch.Generator().manual_seed(3)
M, E = 9920, 1024
if kind == "nt":
    A = torch.randn(M, E, generator=g).to(dev).bfloat16(); B = (torch.randn(3 * E, E, generator=g) / 32).to(dev).bfloat16()
    out = torch.zeros((M, 3 * E), dtype=torch.bfloat16, device=dev); bias = torch.zeros(3 * E, device=dev)
    for _ in range(10):
        rt.gemm(L.EPI_STORE_T, A, B, M, 3 * E, E, out, 3 * E, bias=bias)
else:
    dY = torch.randn(M, 3 * E, generator=g).to(dev).bfloat16(); X = torch.randn(M, E, generator=g).to(dev).bfloat16()
    dW = torch.zeros((3 * E, E), device=dev); db = torch.zeros(3 * E, device=dev)
    for _ in range(10):
        rt.wgrad(dY, 3 * E, X, E, M, dW, db)
torch.cuda.synchronize()
