"""LayerNorm kernels at the production shape (9920 x 1024): forward, backward with / without the
dropout mask on the operand copy - how much of each launch is Philox."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime, _stream
from tim_amd._lib import call, ptr
dev = "cuda:0"; rt = Runtime("fp16"); g = torch.Generator().manual_seed(3)
M, E, FF = 9920, 1024, 2048
# several buffer sets so that consecutive launches do not find their input in the caches
sets = []
for i in range(6):
    y = torch.randn(M, E, generator=g).to(dev); dx = torch.randn(M, E, generator=g).to(dev)
    sets.append((y, dx, torch.empty((M, E), device=dev), torch.empty((M, E), dtype=torch.float16, device=dev),
                 torch.empty((M, 2), device=dev), torch.empty((M, FF // 8), dtype=torch.uint8, device=dev)))
w = torch.ones(E, device=dev); b = torch.zeros(E, device=dev); dg = torch.zeros(E, device=dev); db = torch.zeros(E, device=dev)
def t(fn, n=30):
    for i in range(6): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def fwd(p):
    def f(i):
        y, dx, of, ot, st, mk = sets[i % 6]
        call("timhip_layernorm_fwd", rt.prec, ptr(y), M, E, E, 0, ptr(w), ptr(b), None, 0, ptr(ot), E, ptr(st), _stream())
    return f
def bwd(p):
    def f(i):
        y, dx, of, ot, st, mk = sets[i % 6]
        call("timhip_layernorm_bwd", rt.prec, ptr(dx), E, ptr(y), E, ptr(st), M, E, 0, ptr(w), ptr(of), E, ptr(ot), E,
             p, 7, 17, ptr(dg), ptr(db), None, _stream())
    return f
def cast(i):
    y, dx, of, ot, st, mk = sets[i % 6]
    call("timhip_cast_rows", rt.prec, ptr(y), M, E, E, ptr(ot), E, 0.0, 0, 0, None, _stream())
def copy(i):
    y, dx, of, ot, st, mk = sets[i % 6]
    of.copy_(y)
t(fwd(0.0))
print("reference points: fp32 -> fp16 cast of the same matrix (40 MB in, 20 MB out) %.1f us | torch fp32 copy (40 + 40 MB) %.1f us" % (t(cast), t(copy)))
print("ln_fwd (operand copy + statistics, no keep-bits) %.1f us" % t(fwd(0.0)))
print("ln_bwd  no mask %.1f us | with dropout mask %.1f us" % (t(bwd(0.0)), t(bwd(0.1))))
def bwd_part(which):
    def f(i):
        y, dx, of, ot, st, mk = sets[i % 6]
        call("timhip_layernorm_bwd", rt.prec, ptr(dx), E, ptr(y), E, ptr(st), M, E, 0, ptr(w), ptr(of) if which != "t" else None, E,
             ptr(ot) if which != "f" else None, E, 0.0, 7, 17, ptr(dg), ptr(db), None, _stream())
    return f
print("ln_bwd  fp32 output only (80 MB in, 40 out) %.1f us | operand copy only (80 in, 20 out) %.1f us" % (t(bwd_part("f")), t(bwd_part("t"))))
