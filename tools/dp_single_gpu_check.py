"""Exercise the multi-rank code path of tim_amd.dp.DataParallel on ONE GPU: a 1-rank RCCL group with the exchange forced on
(force=True): every bucket goes through the comm-stream narrow -> all-to-all -> fp32 sum -> all-gather -> widen sequence (each
collective is a copy on one rank), concurrently with the rest of the backward.  Checks that the gradients come back as the
bf16 rounding of a plain run's, and measures what the side-stream work costs the data chain (step time with vs without)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
from tim_amd.config import named_config
from tim_amd.dp import DataParallel
cfg = named_config("C2a"); dev = torch.device("cuda", 0)
B = int(os.environ.get("DP_CHECK_BATCH", "8"))
model, _ = bench.build_model(cfg, "bf16", dev); model.train()
dp = DataParallel(model, force=True, buckets_per_exchange=int(os.environ.get("DP_GROUP", "3")))
assert dp.active and dp.world == 1
batch = bench.make_batch(cfg, B, 15, 10, 100, dev); R = [None]
model.rt.step = 0
bench.step_fn(dp, batch, 15, 10, R); torch.cuda.synchronize()
g1 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
m2, _ = bench.build_model(cfg, "bf16", dev); m2.train(); m2.rt.step = 0
bench.step_fn(m2, batch, 15, 10, R); torch.cuda.synchronize()
bad = 0
for n, p in m2.named_parameters():
    if p.grad is None: continue
    a, b = g1[n], p.grad
    # one rank: the mean is the value itself, rounded once to the wire dtype (half a bf16 ulp; the LayerNorm / token gradients
    # are summed with atomics, so the two runs may differ in the last fp32 bits BEFORE that rounding: allow a whole ulp)
    if not bool(((a - b).abs() <= 2.0 ** -8 * b.abs() + 2e-3 * b.abs().max()).all()):
        bad += 1; print("MISMATCH", n, (a - b).abs().max().item(), b.abs().max().item())
print("RESULT params", len(g1), "mismatches", bad)


def timed(mdl, n=20):
    for _ in range(3): bench.step_fn(mdl, batch, 15, 10, R)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): bench.step_fn(mdl, batch, 15, 10, R)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for Bi in (64,):
    batch = bench.make_batch(cfg, Bi, 15, 10, 100, dev); R = [None]
    t_plain = timed(m2)
    t_dp = timed(dp)
    dp.begin_step_timing(); bench.step_fn(dp, batch, 15, 10, R); comm_ms, nbytes = dp.end_step_timing()
    print("INTERFERENCE B=%d: step %.3f ms plain, %.3f ms with the exchange kernels on the comm stream (+%.1f %%); comm stream busy "
          "%.3f ms per step" % (Bi, t_plain, t_dp, (t_dp / t_plain - 1) * 100, comm_ms))
dist.destroy_process_group()
