import sys; sys.path.insert(0, "/root/repo")
import torch
from tim_amd.functional import Runtime
rt = Runtime("bf16"); dev = "cuda:0"; g = torch.Generator().manual_seed(3)
M, E, FF = 9920, 1024, 2048
items = []
for (no, ko) in ((E, FF), (FF, E), (E, E), (3 * E, E)):
    dY = torch.randn(M, no, generator=g).to(dev).bfloat16(); X = torch.randn(M, ko, generator=g).to(dev).bfloat16()
    items.append((dY, no, X, ko, torch.zeros((no, ko), device=dev), torch.zeros(no, device=dev)))
for _ in range(3): rt.wgrad_group(items, M, accumulate=False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): rt.wgrad_group(items, M, accumulate=False)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("grouped wgrad %.1f us  %.0f TF" % (ms * 1e3, 2.0 * M * 12 * E * E / ms / 1e9))
