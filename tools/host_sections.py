"""Where does the host spend an eager C2a step?  Wraps tim_amd._lib.call and torch.empty with timers and times the two
autograd Functions' forward / backward bodies (the backward runs in the autograd engine's thread: cProfile does not see it)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tim_amd import _lib as L, functional as F
from tim_amd.config import named_config
cfg = named_config("C2a"); dev = torch.device("cuda", 0)
model, _ = bench.build_model(cfg, "fp16", dev); model.train()
batch = bench.make_batch(cfg, 64, 15, 10, 100, dev); R = [None]
for _ in range(40): bench.step_fn(model, batch, 15, 10, R)
torch.cuda.synchronize()
T = collections.Counter(); N = collections.Counter()
def wrap(obj, name, key):
    f = getattr(obj, name)
    is_static = isinstance(getattr(obj, "__dict__", {}).get(name), staticmethod)
    def g(*a, **k):
        t0 = time.perf_counter()
        try: return f(*a, **k)
        finally:
            T[key] += time.perf_counter() - t0; N[key] += 1
    setattr(obj, name, staticmethod(g) if is_static else g)
    return f
orig_call = L.call
wrap(L, "call", "ctypes call"); F.call = L.call
import tim_amd.losses as LS; LS.call = L.call
wrap(torch, "empty", "torch.empty"); wrap(torch, "zeros", "torch.zeros")
for cls, nm in ((F.TimeMlpFn, "time_mlp"), (F.EncoderFn, "encoder")):
    for meth in ("forward", "backward"):
        f = getattr(cls, meth)
        def mk(f, key):
            def g(*a, **k):
                t0 = time.perf_counter()
                try: return f(*a, **k)
                finally:
                    T[key] += time.perf_counter() - t0; N[key] += 1
            return staticmethod(g)
        setattr(cls, meth, mk(f, "%s.%s" % (nm, meth)))
wrap(type(model), "_alloc_grad_buckets", "_alloc_grad_buckets")
wrap(type(model), "_layer_params", "_layer_params")
wrap(type(model), "_workspace", "_workspace")
wrap(type(model), "_plan", "_plan")
wrap(F, "_f32c", "_f32c")
wrap(F.Runtime, "weight", "rt.weight")
wrap(F.Runtime, "weight_split", "rt.weight_split")
wrap(F.Runtime, "grad_scale", "rt.grad_scale")
wrap(F.Runtime, "gemm_many", "rt.gemm_many")
wrap(F.Runtime, "wgrad_many", "rt.wgrad_many")
wrap(F.Runtime, "invalidate_weights", "rt.invalidate_weights")
K = 30
t0 = time.perf_counter()
for _ in range(K): bench.step_fn(model, batch, 15, 10, R)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host issue %.3f ms/step (GPU-inclusive %.3f)" % ((t1 - t0) / K * 1e3, (time.perf_counter() - t0) / K * 1e3))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("%-22s %7.3f ms/step  %5.1f calls/step  %6.1f us/call" % (k, v / K * 1e3, N[k] / K, v / N[k] * 1e6))
