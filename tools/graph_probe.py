"""Does capturing the whole fwd+bwd step in a HIP graph pay?  Eager vs graph replay, per workload.
(Probe: the dropout seed is a launch argument, so a replay repeats the masks of the captured step.)
usage: python tools/graph_probe.py [C1 C2a ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tim_amd.config import named_config

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
names = sys.argv[1:] or ["C1", "C2a"]
for name in names:
    cfg = named_config(name)
    nv, na = (10, 0) if name == "C1" else (15, 10)
    B = 64
    model, _ = bench.build_model(cfg, "bf16", dev, seed=0)
    model.train(True)
    batch = bench.make_batch(cfg, B, nv, na, seed=100, dev=dev)
    R = [None]
    for _ in range(5):
        bench.step_fn(model, batch, nv, na, R)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        bench.step_fn(model, batch, nv, na, R)
    t_host = (time.perf_counter() - t0) / 30 * 1e3
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 30 * 1e3
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                bench.step_fn(model, batch, nv, na, R)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            bench.step_fn(model, batch, nv, na, R)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            g.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / 30 * 1e3
    except Exception as e:  # noqa
        import traceback
        traceback.print_exc()
        t_graph = float("nan")
    print("%-4s B=%d eager %.3f ms/step (host enqueue %.3f) ; graph replay %.3f ms/step" % (name, B, t_eager, t_host, t_graph),
          flush=True)
    # the other stream configuration of the backward
    model.rt.overlap_wgrad = not model.rt.overlap_wgrad
    for _ in range(3):
        bench.step_fn(model, batch, nv, na, R)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        bench.step_fn(model, batch, nv, na, R)
    torch.cuda.synchronize()
    t_eager1 = (time.perf_counter() - t0) / 30 * 1e3
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        bench.step_fn(model, batch, nv, na, R)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        bench.step_fn(model, batch, nv, na, R)
    for _ in range(5):
        g1.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        g1.replay()
    torch.cuda.synchronize()
    print("     other stream configuration: eager %.3f ms/step ; graph replay %.3f ms/step" % (t_eager1, (time.perf_counter() - t0) / 30 * 1e3),
          flush=True)
