"""How long does the host need to ENQUEUE a step vs how long does the GPU need to run it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tim_amd.config import named_config
cfg = named_config("C2a"); dev = torch.device("cuda", 0)
model, _ = bench.build_model(cfg, "fp16", dev); model.train()
batch = bench.make_batch(cfg, 64, 15, 10, 100, dev); R = [None]
for _ in range(5): bench.step_fn(model, batch, 15, 10, R)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): bench.step_fn(model, batch, 15, 10, R)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): bench.step_fn(model, batch, 15, 10, R)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
