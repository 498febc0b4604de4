"""Does an eager step cost the host more once a HIP graph of the step exists in the process?  (bench.py times K replays and
then K eager steps: the eager figure must not be an artefact of the capture.)  Tuning tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tim_amd.config import named_config
from tim_amd.graph import GraphedStep
cfg = named_config("C2a"); dev = torch.device("cuda", 0)
model, _ = bench.build_model(cfg, "fp16", dev); model.train()
batch = bench.make_batch(cfg, 64, 15, 10, 100, dev); R = [None]
step = lambda: bench.step_fn(model, batch, 15, 10, R)
def timed(fn, n, tag):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-44s issue %.2f ms/step, total %.2f ms/step" % (tag, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
for _ in range(10): step()
timed(step, 20, "eager, before any capture")
timed(step, 20, "eager, before any capture (again)")
gs = GraphedStep(model, step)
for _ in range(5): gs()
timed(gs, 20, "graph replay")
timed(step, 3, "eager, first 3 steps after the capture")
timed(step, 20, "eager, steps 4-23 after the capture")
timed(step, 20, "eager, steps 24-43 after the capture")
from tim_amd import functional as F
F.graph_safe_dropout(dev, enable=False)
timed(step, 20, "eager, salt off")
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_reserved() >> 20, "MiB reserved")
