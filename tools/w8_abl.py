"""(T) Timing-only ablations of the eight-phase weight-gradient kernel (wgrad_p8_kernel, two C2a layers = 256 tiles): what the bias
MFMAs, the 64-byte gather runs of the X pieces, the result stores, the fragment reads and the LDS-DMA stream each cost.
TIM_AMD_LIB=tim_amd/libtimhip_tuning.so python tools/w8_abl.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import _lib as L
from tim_amd.functional import Runtime
dev = "cuda:0"
rt = Runtime("fp16")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
E, FF, M = 1024, 2048, 9920


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


g = torch.Generator().manual_seed(5)
items, fl = [], 0.0
for h in range(2):
    for no, ko in [(E, FF), (FF, E), (E, E), (3 * E, E)]:
        Y = torch.randn(M, no, generator=g).to(dev).half()
        X = torch.randn(M, ko, generator=g).to(dev).half()
        items.append((Y, no, X, ko, torch.zeros((no, ko), device=dev), torch.zeros(no, device=dev)))
        fl += 2.0 * no * ko * M
nobias = [(Y, no, X, ko, dW, None) for (Y, no, X, ko, dW, db) in items]
arms = [("as shipped", None, items), ("no bias vectors asked for", None, nobias), ("1: no bias MFMAs", "1", items),
        ("2: X pieces in 128-byte runs", "2", items), ("4: no result stores", "4", items), ("7: 1 + 2 + 4", "7", items),
        ("8: fragment reads in step 0 only", "8", items), ("16: no DMA after the prologue", "16", items), ("24: MFMAs + barriers only", "24", items)]
# schedule arms (TIMHIP_W8_SCH; results are right in all of them)
arms += [("SCH 1: phase-1 pieces moved to phase 2", "s1", items), ("SCH 2: no s_setprio", "s2", items), ("SCH 3: 1 + 2", "s3", items),
         ("SCH 4: pieces before the reads", "s4", items), ("SCH 5: 1 + 4", "s5", items), ("SCH 8: reads waited for behind the barrier", "s8", items)]
if len(sys.argv) > 2 and sys.argv[2] == "sch":
    arms = [a for a in arms if a[1] is None and a[2] is items or (a[1] or "").startswith("s")]
res = {a[0]: [] for a in arms}
for r in range(R):
    for name, v, its in arms:
        os.environ.pop("TIMHIP_W8_ABL", None)
        os.environ.pop("TIMHIP_W8_SCH", None)
        if v is not None:
            os.environ["TIMHIP_W8_SCH" if v.startswith("s") else "TIMHIP_W8_ABL"] = v.lstrip("s")
        res[name].append(timeit(lambda: rt.wgrad_group(its, M, accumulate=False)))
os.environ.pop("TIMHIP_W8_ABL", None)
os.environ.pop("TIMHIP_W8_SCH", None)
med = lambda v: sorted(v)[len(v) // 2]
print("box: %s; two layers' weight gradients (%.0f GF), %d rounds x 10, us = median (min)" % (torch.cuda.get_device_name(0), fl / 1e9, R))
for name, v, its in arms:
    t = res[name]
    print("  %-36s %7.1f (%7.1f) us = %5.0f TF" % (name, med(t), min(t), fl / med(t) / 1e6))
