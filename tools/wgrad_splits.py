"""Split count sweep of the TN weight-gradient kernel (tuning build: TIMHIP_WGRAD_SPLITS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd.functional import Runtime
dev = "cuda:0"; rt = Runtime("bf16"); g = torch.Generator().manual_seed(3)
M, E, FF = 9920, 1024, 2048
for name, Nout, Kout in (("in_proj", 3 * E, E), ("out_proj", E, E), ("ffn1", FF, E), ("ffn2", E, FF)):
    dY = torch.randn(M, Nout, generator=g).to(dev).bfloat16(); X = torch.randn(M, Kout, generator=g).to(dev).bfloat16()
    dW = torch.zeros((Nout, Kout), device=dev); db = torch.zeros(Nout, device=dev)
    row = []
    for sk in ("", "1", "2", "3", "4", "5", "6", "8", "10", "12", "16"):
        if sk: os.environ["TIMHIP_WGRAD_SPLITS"] = sk
        else: os.environ.pop("TIMHIP_WGRAD_SPLITS", None)
        for _ in range(3): rt.wgrad(dY, Nout, X, Kout, M, dW, db)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): rt.wgrad(dY, Nout, X, Kout, M, dW, db)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        row.append("sk%s %.1fus %.0fTF" % (sk or "dflt", ms * 1e3, 2.0 * M * Nout * Kout / ms / 1e9))
    print(name, " | ".join(row), flush=True)
