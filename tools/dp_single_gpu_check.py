"""Exercise the multi-rank code path of tim_amd.dp.DataParallel on ONE GPU: a 1-rank RCCL group with the
wrapper told world=2, so every bucket goes through comm-stream all-reduce (identity) and /2: the gradients
must come out exactly half of a plain single-rank run with the same dropout seed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
from tim_amd.config import named_config
from tim_amd.dp import DataParallel
cfg = named_config("C2a"); dev = torch.device("cuda", 0)
model, _ = bench.build_model(cfg, "bf16", dev); model.train()
dp = DataParallel(model); dp.world = 2
for p in dp._small: dp._hook_handles.append(p.register_post_accumulate_grad_hook(dp._on_small_grad))
batch = bench.make_batch(cfg, 8, 15, 10, 100, dev); R = [None]
model.rt.step = 0
bench.step_fn(dp, batch, 15, 10, R); torch.cuda.synchronize()
g1 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
m2, _ = bench.build_model(cfg, "bf16", dev); m2.train(); m2.rt.step = 0
bench.step_fn(m2, batch, 15, 10, R); torch.cuda.synchronize()
bad = 0
for n, p in m2.named_parameters():
    if p.grad is None: continue
    a, b = g1[n], p.grad
    if not torch.allclose(a, b * 0.5, rtol=1e-3, atol=1e-4 * b.abs().max().item() + 1e-12):
        bad += 1; print("MISMATCH", n, (a - 0.5 * b).abs().max().item(), b.abs().max().item())
print("RESULT params", len(g1), "mismatches", bad)
dist.destroy_process_group()
