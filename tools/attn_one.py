"""Run the MFMA attention fwd / bwd kernels alone at the C2a shape (timing, rocprofv3 PMC, per-phase clocks).

    python tools/attn_one.py [p_drop]
    ATT_PREC=3         fp16 operands (default 0: bf16)
    ATT_SETS=8         second timing with the operands out of HBM: that many buffer sets in rotation (162 MB each)
    TIMHIP_ATTN_KS=0   the fused backward with one wave per row block instead of the pipeline (TIMHIP_ATTN_FUSED=0: two kernels)
  with TIM_AMD_LIB=tim_amd/libtimhip_tuning.so (make -C tim_amd/csrc TUNING=1), ATT_ABL = sum of
    1 no scratch stores, 2 no dqkv stores, 4 operand rows from two cached rows (no DRAM traffic),
    16 shader-clock stamps per block and phase (printed below), 32 / 64 / 128 drop the dQ units / dS units / sweeps 1.. of the pipeline
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
B, S, F, H, Dh = 64, 155, 100, 8, 128
E = H * Dh; dev = "cuda:0"
g = torch.Generator().manual_seed(1)
PREC = int(os.environ.get("ATT_PREC", "0"))   # 0 bf16, 3 fp16
HD = torch.float16 if PREC == 3 else torch.bfloat16
qkv = torch.randn(B * S, 3 * E, generator=g).to(dev).to(HD)
do = torch.randn(B * S, E, generator=g).to(dev).to(HD)
o = torch.zeros((B * S, E), dtype=HD, device=dev); lse = torch.empty((B, H, S), device=dev)
dqkv = torch.zeros_like(qkv)
desc = L.TimDesc(B, S, F, E // 2, E, H, 2 * E, PREC, p, 99, 1, int(os.environ.get("ATT_ABL", "0")) << 8)
wsb = max(L.load().timhip_attention_bwd_workspace_bytes(C.byref(desc)), B * H * 64); ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
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
NSETS = int(os.environ.get("ATT_SETS", "0"))
if NSETS:   # operands out of HBM: rotate through NSETS copies of every buffer (162 MB each; the infinity cache holds 256 MB)
    sets = [(qkv.clone(), o.clone(), lse.clone(), do.clone(), torch.zeros_like(qkv)) for _ in range(NSETS)]
    def bwd_i(i):
        q_, o_, l_, d_, g_ = sets[i % NSETS]
        L.call("timhip_attention_bwd", C.byref(desc), L.ptr(q_), L.ptr(o_), L.ptr(l_), L.ptr(d_), L.ptr(g_), L.ptr(ws), wsb, st)
    def fwd_i(i):
        q_, o_, l_, d_, g_ = sets[i % NSETS]
        L.call("timhip_attention_fwd", C.byref(desc), L.ptr(q_), L.ptr(o_), L.ptr(l_), st)
    for f, name in ((fwd_i, "fwd"), (bwd_i, "bwd")):
        for i in range(NSETS): f(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(4 * NSETS): f(i)
        e1.record(); torch.cuda.synchronize()
        print(name, "p=%.2f" % p, "%.1f us  (operands from HBM: %d buffer sets in rotation)" % (e0.elapsed_time(e1) / (4 * NSETS) * 1e3, NSETS))
if int(os.environ.get("ATT_ABL", "0")) & 16:   # tuning build: wave 0's shader-clock stamps of every block (attention_bwd2.hip ATT_STAMP)
    t = ws[:B * H * 64].view(torch.int64).view(B * H, 8).cpu().double()
    names = ["requests + K/V -> LDS", "sweep 0 -> LDS + barrier", "pipeline steps (1a | 1b | next sweep)", "(to phase 2)", "Q load + barrier", "X store + dO load + barrier (to prod 1)", "rest of phase 2"]
    d = t[:, 1:] - t[:, :-1]
    print("per block, wave 0, shader clocks (mean / max over %d blocks); total %.0f" % (B * H, (t[:, 7] - t[:, 0]).mean()))
    for i, n in enumerate(names):
        print("  %-44s %8.0f %8.0f" % (n, d[:, i].mean(), d[:, i].max()))
    first = t[:, 0].min(); print("  first block start -> last block end: %.0f clocks; blocks starting in the first 10%%: %d" % (
        t[:, 7].max() - first, int((t[:, 0] < first + 0.1 * (t[:, 7].max() - first)).sum())))
