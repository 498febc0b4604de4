"""Device-side sliding-window batch assembly at the C2a shapes (B = 64 windows x 50 features x (1024 + 2304) floats):
HIP gather from an HBM-resident feature store vs the reference-style host path (numpy fancy-index gather + host-to-device
copy of the batch).  Timing tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tim_amd.data import DeviceWindowDataset
dev = "cuda:0"
nvid, nfeat, nf, Cv, Ca, B = 24, 3000, 50, 1024, 2304, 64
g = torch.Generator(device=dev).manual_seed(0)
vids = ["v%02d" % i for i in range(nvid)]
vf = {v: torch.randn(nfeat, 1, Cv, device=dev, generator=g) for v in vids}
af = {v: torch.randn(nfeat, 1, Ca, device=dev, generator=g) for v in vids}
st = torch.arange(nfeat).float() * 0.2
ft = {v: torch.stack([st, st + 1.0], 1) for v in vids}
rs = np.random.RandomState(0)
windows = []
for i in range(4000):
    first = int(rs.randint(0, nfeat - 2 * nf))
    windows.append({"video_id": vids[i % nvid], "start_sec": first * 0.2, "feat_indices": np.arange(first, first + 2 * nf, 2),
                    "v_queries": np.zeros((0, 2), np.float32), "v_labels": np.zeros((0, 4), np.int64), "v_action_ids": np.zeros(0, np.int64),
                    "v_narration_ids": [], "a_queries": np.zeros((0, 2), np.float32), "a_labels": np.zeros((0, 4), np.int64),
                    "a_action_ids": np.zeros(0, np.int64), "a_narration_ids": []})
ds = DeviceWindowDataset(windows, nf, nf * 0.4, 15, 10, "audio_visual", vf, ft, af, ft, device=dev)
idx = torch.from_numpy(rs.randint(0, len(windows), B))
for _ in range(3): out = ds.batch(idx)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
for _ in range(50): out = ds.batch(idx)
e1.record(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 50
gpu_ms = e0.elapsed_time(e1) / 50
bytes_alg = 2.0 * B * nf * (Cv + Ca) * 4
print("device batch(): %.1f us GPU time per batch, %.1f us host wall; gather moves %.1f MB algorithmic (read+write) -> %.2f TB/s "
      "if it were the only work" % (gpu_ms * 1e3, wall * 1e6, bytes_alg / 1e6, bytes_alg / (gpu_ms * 1e-3) / 1e12))
from tim_amd import _lib as L
from tim_amd._lib import ptr
win = idx.to(device=dev, dtype=torch.int32); aug = torch.zeros((B, nf), dtype=torch.int32, device=dev)
ov = torch.empty((B, nf, Cv), device=dev); oa = torch.empty((B, nf, Ca), device=dev)
stream = torch.cuda.current_stream().cuda_stream
def kernels():
    L.call("timhip_window_gather", ptr(ds.v["feats"]), Cv, 1, ptr(ds.v_row0), ptr(ds.feat_indices), nf, ptr(win), B, ptr(aug), ptr(ov), stream)
    L.call("timhip_window_gather", ptr(ds.a["feats"]), Ca, 1, ptr(ds.a_row0), ptr(ds.feat_indices), nf, ptr(win), B, ptr(aug), ptr(oa), stream)
for _ in range(5): kernels()
torch.cuda.synchronize(); e0.record()
for _ in range(200): kernels()
e1.record(); torch.cuda.synchronize()
k_us = e0.elapsed_time(e1) / 200 * 1e3
print("the two gather launches alone: %.1f us per batch -> %.2f TB/s of %.1f MB (HBM roofline 8 TB/s: frac %.2f)"
      % (k_us, bytes_alg / (k_us * 1e-6) / 1e12, bytes_alg / 1e6, bytes_alg / (k_us * 1e-6) / 8e12))
# host path: numpy fancy-index gather (what a DataLoader worker does) + pinned H2D copy of the batch
vh = {v: t.cpu().numpy() for v, t in list(vf.items())[:4]}; ah = {v: t.cpu().numpy() for v, t in list(af.items())[:4]}
wl = [w for w in windows if w["video_id"] in vh][:B]
t0 = time.perf_counter()
for _ in range(5):
    vb = np.stack([vh[w["video_id"]][w["feat_indices"], 0] for w in wl]); ab = np.stack([ah[w["video_id"]][w["feat_indices"], 0] for w in wl])
    tv = torch.from_numpy(vb).pin_memory().to(dev, non_blocking=True); ta = torch.from_numpy(ab).pin_memory().to(dev, non_blocking=True)
    torch.cuda.synchronize()
host = (time.perf_counter() - t0) / 5
print("host path (1 worker: numpy gather + pin + H2D): %.2f ms per batch -> x%.0f" % (host * 1e3, host / wall))
