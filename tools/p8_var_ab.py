"""Experiment arms of the eight-phase NT main loop (gemm_nt_p8_kernel<.., VAR>), interleaved, plain 16-bit stores, fp16:
VAR bit 0 = no s_setprio around the MFMA segments, bit 1 = fragment reads waited for after the phase's first barrier.
Needs the experiment build:  (cd tim_amd/csrc && mkdir -p build_p8v && cp build/*.o build_p8v/ && hipcc --offload-arch=gfx950 -O3
-std=c++17 -fPIC -DTIMHIP_P8_VARIANTS -c gemm_pp.hip -o build_p8v/gemm_pp.o && hipcc --offload-arch=gfx950 -shared -fPIC -o
../libtimhip_p8v.so build_p8v/*.o);   TIM_AMD_LIB=tim_amd/libtimhip_p8v.so python tools/p8_var_ab.py > profiles/r06_f_p8_variants_ab.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime
dev = "cuda:0"
rt = Runtime("fp16")
M, E, FF = 9920, 1024, 2048
g = torch.Generator().manual_seed(3)


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


print("lib:", L.LIB_PATH)
for name, N, K, tm in (("in_proj fwd", 3 * E, E, "8"), ("ffn1 fwd / ffn2 dgrad", FF, E, "10")):
    A = torch.randn(M, K, generator=g).to(dev).half()
    B = (torch.randn(N, K, generator=g) / 32).to(dev).half()
    oT = torch.zeros((M, N), dtype=torch.float16, device=dev)
    bias = torch.zeros(N, device=dev)
    os.environ["TIMHIP_GEMM_P8"] = tm
    L.reload_env()
    res = {v: [] for v in "0123"}
    ref = None
    for r in range(6):
        for v in "0123":
            os.environ["TIMHIP_GEMM_P8_VAR"] = v
            res[v].append(timeit(lambda: rt.gemm(L.EPI_STORE_T, A, B, M, N, K, oT, N, bias=bias)))
            if r == 0:
                cur = oT.float().clone()
                if ref is None:
                    ref = cur
                else:
                    print("   var %s max |diff| vs var 0: %.3g" % (v, (cur - ref).abs().max().item()))
    fl = 2.0 * M * N * K
    med = lambda x: sorted(x)[len(x) // 2]
    print("%s  %d x %d x %d  tile %s0 rows:" % (name, M, N, K, {"8": "256 = 8 x 32, TM 8: ", "10": "320 = 10 x 32, TM 10: "}[tm][:3]) +
          "   ".join("var %s %.1f (%.1f) us %4.0f TF" % (v, med(res[v]), min(res[v]), fl / med(res[v]) / 1e6) for v in "0123"), flush=True)
