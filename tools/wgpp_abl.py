"""Timing-only ablations of the ping-pong weight-gradient kernel (wgrad_pp.hip, tuning build): the layer's grouped launch with
the fragment reads / the MFMAs / the DMA pieces removed (TIMHIP_WGPP_ABL bits 1 / 2 / 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd.functional import Runtime
dev = "cuda:0"; rt = Runtime("fp16"); g = torch.Generator().manual_seed(3)
M, E, FF = 9920, 1024, 2048
items = []
for (no, ko) in ((E, FF), (FF, E), (E, E), (3 * E, E)):
    dY = torch.randn(M, no, generator=g).to(dev).half(); Xa = torch.randn(M, ko, generator=g).to(dev).half()
    items.append((dY, no, Xa, ko, torch.zeros((no, ko), device=dev), torch.zeros(no, device=dev)))
fl = sum(2.0 * M * it[1] * it[3] for it in items)
ref = [(it[0].float().t() @ it[2].float(), it[0].float().sum(0)) for it in items]
SEQ = [(m, 0) for _ in range(3) for m in (0, 1, 2, 4)] + [(2, 2), (2, 4), (2, 6), (4, 2), (4, 4), (4, 6)]   # A/B/C interleaved three times, then the ablations
for mode, abl in SEQ:
    os.environ["TIMHIP_WGPP_ABL"] = str(abl); os.environ["TIMHIP_WGPP_MODE"] = str(mode)
    for _ in range(3): rt.wgrad_group(items, M, accumulate=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): rt.wgrad_group(items, M, accumulate=False)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    if abl == 0:
        err = max(float((it[4] - r[0]).abs().max() / r[0].abs().max()) for it, r in zip(items, ref))
        errb = max(float((it[5] - r[1]).abs().max() / r[1].abs().max()) for it, r in zip(items, ref))
        print("   mode %d: max rel err dW %.2e db %.2e" % (mode, err, errb))
    print("mode %d (0 ping-pong, 1 free-running, 2 merged-phase ping-pong, 3 = 2 with the DMA in the LOAD phase, 4 = 8 consumers + 4 loader waves) ABL %d (%s%s%s): %.1f us (%.0f TF-equivalent)" % (mode, abl, "no-reads " if abl & 1 else "", "no-mfma " if abl & 2 else "",
                                                            "no-dma" if abl & 4 else "", us, fl / us / 1e6), flush=True)
