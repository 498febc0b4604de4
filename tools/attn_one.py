"""Run the MFMA attention fwd/bwd kernels alone (timing + rocprofv3 PMC)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
B, S, F, H, Dh = 64, 155, 100, 8, 128
E = H * Dh; dev = "cuda:0"
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B * S, 3 * E, generator=g).to(dev).bfloat16()
do = torch.randn(B * S, E, generator=g).to(dev).bfloat16()
o = torch.zeros((B * S, E), dtype=torch.bfloat16, device=dev); lse = torch.empty((B, H, S), device=dev)
dqkv = torch.zeros_like(qkv)
desc = L.TimDesc(B, S, F, E // 2, E, H, 2 * E, 0, p, 99, 1, int(os.environ.get("ATT_ABL", "0")) << 8)
wsb = L.load().timhip_attention_bwd_workspace_bytes(C.byref(desc)); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def fwd(): L.call("timhip_attention_fwd", C.byref(desc), L.ptr(qkv), L.ptr(o), L.ptr(lse), st)
def bwd(): L.call("timhip_attention_bwd", C.byref(desc), L.ptr(qkv), L.ptr(o), L.ptr(lse), L.ptr(do), L.ptr(dqkv), L.ptr(ws), wsb, st)
for f, name in ((fwd, "fwd"), (bwd, "bwd")):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(name, "p=%.2f" % p, "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
