"""The weight gradients of two C2a encoder layers: ONE launch of 256 eight-phase 256 x 256 tiles (wgrad_p8_kernel) against the
same eight products as two rounds of 128 x 256 tiles (wgrad_ld_kernel, TIMHIP_WGRAD_P8=0) and against two single-layer launches;
interleaved, median (min) of R rounds.   python tools/wg_pair_ab.py [rounds] > profiles/r06_o_wgrad_pair_ab.txt"""
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


def setp8(v):
    os.environ["TIMHIP_WGRAD_P8"] = v
    L.reload_env()


g = torch.Generator().manual_seed(5)
items, fl = [], 0.0
for h in range(2):
    for no, ko in [(E, FF), (FF, E), (E, E), (3 * E, E)]:
        Y = torch.randn(M, no, generator=g).to(dev).half()
        X = torch.randn(M, ko, generator=g).to(dev).half()
        items.append((Y, no, X, ko, torch.zeros((no, ko), device=dev), torch.zeros(no, device=dev)))
        fl += 2.0 * no * ko * M
res = {"p8": [], "p8_2": [], "ld2": [], "ld1": []}
for r in range(R):
    setp8("1")
    res["p8"].append(timeit(lambda: rt.wgrad_group(items, M, accumulate=False)))
    os.environ["TIMHIP_WGRAD_P8_PH"] = "2"
    L.reload_env()
    res["p8_2"].append(timeit(lambda: rt.wgrad_group(items, M, accumulate=False)))
    os.environ["TIMHIP_WGRAD_P8_PH"] = "4"
    L.reload_env()
    setp8("0")
    res["ld2"].append(timeit(lambda: rt.wgrad_group(items, M, accumulate=False)))
    res["ld1"].append(timeit(lambda: (rt.wgrad_group(items[:4], M, accumulate=False), rt.wgrad_group(items[4:], M, accumulate=False))))
setp8("1")
med = lambda v: sorted(v)[len(v) // 2]
print("box: %s; two layers' weight gradients (%.0f GF), %d rounds x 10, us = median (min)" % (torch.cuda.get_device_name(0), fl / 1e9, R))
for k, name in [("p8", "one launch, 256 eight-phase tiles of 256 x 256"), ("p8_2", "the same, two 32-MFMA phases per step"), ("ld2", "one launch, 512 tiles of 128 x 256 (two rounds)"),
                ("ld1", "two launches of 256 tiles of 128 x 256")]:
    t = res[k]
    print("  %-52s %7.1f (%7.1f) us = %5.0f TF" % (name, med(t), min(t), fl / med(t) / 1e6))
