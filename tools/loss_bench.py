"""Loss tail at the C2a shapes (B = 64: 960 visual query rows x 3806 action classes, 97 verbs, 300 nouns; 640 audio rows x 44;
DRLoc 64 x 32 pairs): fused HIP kernels vs the stock torch ops the reference's train.py issues.  Timing tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tim_amd import losses
from tim_amd.config import named_config
from tim_amd.tim import TIM
dev = "cuda:0"
def timeit(f, n=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
for rows, C in ((960, 3806), (960, 300), (960, 97), (640, 44)):
    x = torch.randn(rows, C, generator=g).to(dev).requires_grad_(True)
    ya = torch.randint(-1, C, (rows,), generator=g).to(dev); yb = torch.randint(-1, C, (rows,), generator=g).to(dev)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.2, ignore_index=-1)
    def ref():
        x.grad = None
        va, vb = ya != -1, yb != -1
        l = 0.3 * crit(x[va], ya[va]).mean() + 0.7 * crit(x[vb], yb[vb]).mean()
        l.backward()
    def ours():
        x.grad = None
        losses.mixup_cross_entropy(x, ya, yb, 0.3, 0.2).backward()
    print("mixup CE fwd+bwd [%d x %d]: torch ops %.1f us, fused %.1f us" % (rows, C, timeit(ref), timeit(ours)), flush=True)
cfg = named_config("C2a")
m = TIM(cfg.num_class, d_model=cfg.d_model, num_layers=1, input_modality="audio_visual", data_modality="audio_visual",
        num_feats=cfg.num_feats, precision="bf16").to(dev)
feats = torch.randn(64, cfg.F, cfg.E, generator=g).to(dev).requires_grad_(True)
l = cfg.num_feats
def ours():
    feats.grad = None
    losses.dense_relative_localization_loss_crossmodal(feats[:, :l], feats[:, l:], m, 32).backward()
ref_mlp = torch.nn.Sequential(torch.nn.Linear(4 * cfg.d_model, cfg.d_model), torch.nn.ReLU(), torch.nn.Linear(cfg.d_model, cfg.d_model),
                              torch.nn.ReLU(), torch.nn.Linear(cfg.d_model, 1)).to(dev)
def ref():
    feats.grad = None
    n = 64
    p1, p2 = losses.position_sampling(l, 32, n)
    d = torch.abs((p1 - p2).float()).to(dev) / l
    a = losses.collect_samples(feats[:, :l], p1.to(dev), n).transpose(1, 2)
    b = losses.collect_samples(feats[:, l:], p2.to(dev), n).transpose(1, 2)
    torch.nn.functional.l1_loss(d, ref_mlp(torch.cat([a, b], dim=2)).squeeze(2)).backward()
print("DRLoc crossmodal fwd+bwd [64 x 32 pairs]: torch ops (fp32) %.1f us, HIP path (bf16) %.1f us" % (timeit(ref), timeit(ours)), flush=True)
