"""A complete TIM recognition training loop on the MI355X path, on synthetic data - the shape of
recognition/scripts/train.py:190-366 with every piece served by tim_amd:

    DeviceWindowDataset.batch()      <- DataLoader workers + default_collate + .cuda()      (sliding_window.py:341-421)
    model(times, "time_mlp"), model(inputs, "encoder", ...)                                  (models/tim.py)
    mixup of the inputs              <- utils/mixup.py:4-22 (three lerps; stays torch ops on the device)
    losses.mixup_cross_entropy       <- criterion + mixup_criterion per head                (train.py:218-316)
    losses.dense_relative_localization_loss_crossmodal                                       (train.py:318-349)
    torch.optim.AdamW                                                                        (train.py:66-70)

    python examples/train_synthetic.py [--steps 20] [--config tiny|C2a] [--batch 8]
    python -m torch.distributed.run --nproc-per-node N examples/train_synthetic.py ...      (data parallel over RCCL)
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tim_amd import losses, synth  # noqa: E402
from tim_amd.config import named_config  # noqa: E402
from tim_amd.data import DeviceWindowDataset  # noqa: E402
from tim_amd.tim import TIM  # noqa: E402


def synthetic_dataset(cfg, n_videos, n_windows, nv, na, dev, seed=0):
    """in-memory tables in the reference dataset's format (see tests/golden/batch_inputs.py)"""
    rs = np.random.RandomState(seed)
    nf = cfg.num_feats
    vids = ["vid%02d" % i for i in range(n_videos)]
    n_feat = 4 * nf
    st = (np.arange(n_feat) * 0.2).astype(np.float32)
    ft = np.stack([st, st + 1.0], 1)
    vf = {v: synth.normal(seed, "v" + v, (n_feat, 2, cfg.visual_input_dim)).astype(np.float32) for v in vids}
    af = {v: synth.normal(seed, "a" + v, (n_feat, 2, cfg.audio_input_dim)).astype(np.float32) for v in vids}
    vc, ac = cfg.num_class[0], cfg.num_class[1]
    windows = []
    for i in range(n_windows):
        first = int(rs.randint(0, n_feat - 2 * nf))
        start = first * 0.2
        q = lambda m: (start + np.sort(rs.rand(m, 2) * nf * 0.4, axis=1)).astype(np.float32)
        kv, ka = int(rs.randint(1, nv + 1)), int(rs.randint(1, na + 1))
        vl = np.stack([rs.randint(0, vc[0], kv), rs.randint(0, vc[1], kv), rs.randint(0, vc[2], kv), np.full(kv, -1)], 1)
        al = np.stack([np.full(ka, -1)] * 3 + [rs.randint(0, ac, ka)], 1)
        windows.append({"video_id": vids[i % n_videos], "start_sec": start, "feat_indices": np.arange(first, first + 2 * nf, 2),
                        "v_queries": q(kv), "v_labels": vl.astype(np.int64), "v_action_ids": np.arange(kv), "v_narration_ids": [""] * kv,
                        "a_queries": q(ka), "a_labels": al.astype(np.int64), "a_action_ids": np.arange(ka), "a_narration_ids": [""] * ka})
    return DeviceWindowDataset(windows, nf, nf * 0.4, nv, na, "audio_visual", vf, {v: ft for v in vids}, af, {v: ft for v in vids}, dev)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="tiny")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--lr", type=float, default=1e-3)
    args = ap.parse_args(argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cfg = named_config(args.config)
    nv, na, nf = (4, 2, cfg.num_feats) if args.config == "tiny" else (15, 10, cfg.num_feats)
    model = TIM(cfg.num_class, visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim, d_model=cfg.d_model,
                nhead=cfg.nhead, num_layers=cfg.num_layers, num_feats=nf, feat_drop=0.1, seq_drop=0.1, enc_dropout=0.1,
                precision=args.precision)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=0).items()})
    model = model.to(dev).train()
    run = model
    if world > 1:
        from tim_amd.dp import DataParallel
        run = DataParallel(model)
    opt = torch.optim.AdamW(model.parameters(), lr=args.lr, weight_decay=1e-4)
    ds = synthetic_dataset(cfg, 4, 64, nv, na, dev, seed=rank)
    g = torch.Generator().manual_seed(rank)
    hist = []
    for step in range(args.steps):
        idx = torch.randint(0, len(ds), (args.batch,), generator=g)
        visual, audio, times, label, _ = ds.batch(idx)
        te = run(times, "time_mlp")
        # mixup of the inputs (utils/mixup.py:4-22)
        lam = float(np.random.RandomState(step).beta(0.2, 0.2))
        perm = torch.randperm(args.batch, generator=g).to(dev)
        visual, audio, te = [lam * t + (1 - lam) * t[perm] for t in (visual, audio, te)]
        tb = {k: v[perm] for k, v in label.items()}
        (verb, noun, action, aud), feats = run([visual, audio], "encoder", te, nv, na)
        ce = lambda x, k: losses.mixup_cross_entropy(x, label[k].reshape(-1), tb[k].reshape(-1), lam, 0.2)
        loss = (ce(verb, "verb") + ce(noun, "noun") + ce(action, "action")) / 3.0 + ce(aud, "class_id")
        loss = loss + 0.3 * losses.dense_relative_localization_loss_crossmodal(feats[:, :nf], feats[:, nf:], model, 8)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        # non-finite guard (what GradScaler.step does in the reference loop, train.py:355-363): the fp16 mode's device-side
        # gradient scale never skips a step by itself, so the loop does - the clip already computes the norm
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        inner = model.module if hasattr(model, "module") else model
        # (two equivalent tests: the norm the clip computed, and the word the weight-gradient kernels OR when they write inf / nan)
        if torch.isfinite(gnorm) and inner.rt.grads_finite():
            opt.step()
        hist.append(loss.item())
        if rank == 0 and (step % 5 == 0 or step == args.steps - 1):
            print("step %3d  loss %.4f" % (step, hist[-1]), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return hist


if __name__ == "__main__":
    main()
