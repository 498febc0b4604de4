"""Batched soft-NMS throughput: one timhip_softnms_1d call over all (video, class) groups of a synthetic evaluation vs the
C oracle (the reference's algorithm, one core) on a sample of the same groups.  Timing script (not a pytest module);
it lives under tests/ because it loads the oracle as its CPU baseline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import numpy as np, torch
from oracle import nms_oracle as N
from tim_amd import nms as hnms
from tests.golden.make_golden_nms_inputs import make_segments
rs = np.random.RandomState(0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
sizes = np.clip(np.exp(rs.normal(5.3, 1.0, G)).astype(int), 1, 20000)      # median ~200, long tail
segs, scores, keys = [], [], []
for g, n in enumerate(sizes):
    s, c = make_segments(1000 + g, int(n), False)
    segs.append(s); scores.append(c); keys.append(np.full(n, g, dtype=np.int64))
S, C, K = np.concatenate(segs), np.concatenate(scores), np.concatenate(keys)
St, Ct, Kt = torch.from_numpy(S).cuda(), torch.from_numpy(C).cuda(), torch.from_numpy(K).cuda()
for _ in range(2): out = hnms.grouped_nms(St, Ct, Kt, 0.1, 0.001, sigma=0.4, method=2, nms="soft")
torch.cuda.synchronize(); t0 = time.perf_counter()
out = hnms.grouped_nms(St, Ct, Kt, 0.1, 0.001, sigma=0.4, method=2, nms="soft")
torch.cuda.synchronize(); gpu = time.perf_counter() - t0
samp = rs.choice(G, size=min(G, 300), replace=False)
t0 = time.perf_counter(); kept = 0
for g in samp:
    inds, _ = N.softnms_1d(segs[g], scores[g], 0.1, 0.4, 0.001, 2); kept += len(inds)
cpu = (time.perf_counter() - t0) * G / len(samp)
print("groups %d, segments %d (max group %d), kept %d: GPU one call %.1f ms (%.2f M segments/s); C oracle, 1 core, extrapolated "
      "from %d groups: %.1f ms -> x%.1f" % (G, len(K), sizes.max(), out[1].numel(), gpu * 1e3, len(K) / gpu / 1e6, len(samp), cpu * 1e3, cpu / gpu))
