"""Two data-parallel ranks end to end on ONE GPU: both processes use GPU 0, the gradient buckets are all-reduced over gloo
(RCCL refuses two ranks on one device) on the device buffers, through exactly the hooks and streams of tim_amd/dp.py.  Each
rank runs forward+backward on its own half of a 2B-window batch (no dropout); the averaged gradients must equal half the
gradients of one process run on all 2B windows (the cotangents are per-window, so the loss is a sum over windows).
launched by tests/test_gpu_dp.py:  python -m torch.distributed.run --nproc-per-node 2 tools/dp_two_rank_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
import bench
from tim_amd.config import named_config
from tim_amd.dp import DataParallel
cfg = named_config("C2a")
cfg.feat_drop = cfg.seq_drop = cfg.enc_dropout = 0.0
B, nv, na = 4, 15, 10
full = bench.make_batch(cfg, world * B, nv, na, 100, dev)


def run(model, batch, R):
    for p in model.parameters():
        p.grad = None
    inner = model.module if hasattr(model, "module") else model
    te = model(batch["times"], "time_mlp")
    heads, feats = model([batch["visual"], batch["audio"]], "encoder", te, nv, na)
    outs = [t for t in heads if t is not None] + [feats]
    torch.autograd.backward(outs, R)
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in inner.named_parameters() if p.grad is not None}


# per-window cotangents, generated for the full batch and sliced per rank
model, _ = bench.build_model(cfg, "bf16", dev)
model.train()
with torch.no_grad():
    te = model(full["times"], "time_mlp")
    heads, feats = model([full["visual"], full["audio"]], "encoder", te, nv, na)
g = torch.Generator().manual_seed(7)
outs = [t for t in heads if t is not None] + [feats]
Rfull = [(torch.randn(o.shape, generator=g) * 0.05).to(dev) for o in outs]
nq = [o.shape[0] // (world * B) for o in outs]          # rows per window of every output
shard = {k: v[rank * B:(rank + 1) * B] for k, v in full.items()}
Rs = [r[rank * B * q:(rank + 1) * B * q] for r, q in zip(Rfull, nq)]
dp = DataParallel(model)
assert dp.world == 2
g_dp = run(dp, shard, Rs)
# gradient accumulation: this rank's windows as two microbatches, the first under no_sync(): must equal the one-pass result
for p in model.parameters():
    p.grad = None
h = B // 2
for i, ctx in enumerate((dp.no_sync(), None)):
    mb = {k: v[i * h:(i + 1) * h] for k, v in shard.items()}
    Rm = [r[i * h * q:(i + 1) * h * q] for r, q in zip(Rs, nq)]
    def fb():
        te_ = dp(mb["times"], "time_mlp")
        hd, ft = dp([mb["visual"], mb["audio"]], "encoder", te_, nv, na)
        torch.autograd.backward([t for t in hd if t is not None] + [ft], Rm)
    if ctx is not None:
        with ctx:
            fb()
    else:
        fb()
torch.cuda.synchronize()
acc_bad = 0
for n, p in model.named_parameters():
    if n in g_dp:
        tol = (3e-3 + 2.0 ** -7) * g_dp[n].abs().max().item() + 1e-12
        if (p.grad - g_dp[n]).abs().max().item() > tol:
            acc_bad += 1
            print("ACCUM MISMATCH rank", rank, n, (p.grad - g_dp[n]).abs().max().item(), g_dp[n].abs().max().item())
os.write(1, ("ACCUM rank %d mismatches %d\n" % (rank, acc_bad)).encode())   # (one write: two ranks share the pipe)
# every rank must hold the same averaged gradients
flat = torch.cat([v.reshape(-1) for v in g_dp.values()]).cpu()
other = flat.clone()
dist.broadcast(other, 0)
same = torch.equal(other, flat)
bad = 0
if rank == 0:
    m2, _ = bench.build_model(cfg, "bf16", dev)
    m2.train()
    g_full = run(m2, full, Rfull)
    for n, b in g_full.items():
        a = g_dp[n]
        tol = (3e-3 + 2.0 ** -7) * b.abs().max().item() + 1e-12   # bf16 GEMMs over B vs 2B rows: different split-K partitions /
        # summation order; plus the bf16 wire format of the exchange (each contribution and the mean rounded once)
        if (a - 0.5 * b).abs().max().item() > tol:
            bad += 1
            print("MISMATCH", n, (a - 0.5 * b).abs().max().item(), b.abs().max().item())
    os.write(1, ("RESULT params %d mismatches %d ranks_agree %s\n" % (len(g_dp), bad, same)).encode())
else:
    os.write(1, ("RANK1 ranks_agree %s\n" % same).encode())
dist.barrier()
dist.destroy_process_group()
