"""Run N steps of one secondary bench workload (for rocprofv3): python tools/prof_secondary.py C4 16 20 [--det-train]
python tools/prof_secondary.py C2a 64 30 --rec-train   (the recognition training iteration: bench.py's c2a_train block)
python tools/prof_secondary.py C2a 8 30                (bench.py's c2a_b8 block)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tim_amd.config import named_config
wl, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
det_train = "--det-train" in sys.argv
rec_train = "--rec-train" in sys.argv
dev = torch.device("cuda", 0)
cfg = named_config(wl)
det = getattr(cfg, "variant", "recognition") == "detection"
nv, na = (10, 0) if wl == "C1" else (15, 10)
if det:
    nv, na = 399, 0
m, _ = bench.build_model(cfg, "fp16", dev)
m.train(det_train or not det)
batch = bench.make_batch(cfg, B, 0 if det else nv, na, seed=100, dev=dev)
R = {"target": bench.make_det_targets(cfg, B, 6, 5, dev)} if det_train else [None]
if rec_train:
    R = bench.make_rec_train_state(m, cfg, B, nv, na, dev)
for _ in range(steps):
    bench.step_fn(m, batch, nv, na, R)
torch.cuda.synchronize()
