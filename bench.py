#!/usr/bin/env python3
"""bench.py — interval-queries/sec of the TIM encoder hot path (fwd+bwd) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision fp16|bf16|bf16x3|fp32] [--batch 64]

Workload (BASELINE.json configs[1] = SURVEY.md C2a): EPIC-100 audio-visual recognition,
d_model 512 (E 1024), 6 layers, 8 heads, 50+50 feature tokens, 15 visual + 10 audio interval
queries per window (S = 155), 64 windows per GPU, training mode with the reference dropout
rates (feat .5 / seq .5 / enc .1), synthetic features, random-init weights of the true shapes.
One step = time_mlp + encoder forward, fixed-cotangent loss (no host loss code), full backward
to every parameter gradient, operand-copy refresh of the weights (as after an optimizer step)
and, for N > 1, the gradient exchange over RCCL (tim_amd/dp.py).  Inputs are resident in HBM before the timed region.

The timed mode is `fp16` (fp16 MFMA operands, fp32 accumulation, split operands for the time MLP and the heads): the fastest
mode whose per-query logits stay within north_star's 1e-3 of the fp32 reference arithmetic; that error is measured in this
run (`parity`), and the plain-bf16 mode (faster by a few per cent, 1e-2 off) is reported beside it (`bf16_mode`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tim_amd import synth  # noqa: E402
from tim_amd.config import named_config  # noqa: E402

GFLOP_PER_QUERY = {"C2a": 1.970, "C2b": 2.636, "C3": 1.583, "C1": 0.232, "C4": 0.398}  # BASELINE.md section 4, fwd+bwd
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP32_TFLOPS = 157.3


def build_model(cfg, precision, dev, seed=0):
    if getattr(cfg, "variant", "recognition") == "detection":
        from tim_amd.detection import TIM
    else:
        from tim_amd.tim import TIM
    kw = dict(visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim, feat_drop=cfg.feat_drop,
              seq_drop=cfg.seq_drop, d_model=cfg.d_model, nhead=cfg.nhead, num_layers=cfg.num_layers,
              enc_dropout=cfg.enc_dropout, input_modality=cfg.input_modality, data_modality=cfg.data_modality,
              num_feats=cfg.num_feats, include_verb_noun=cfg.include_verb_noun, precision=precision)
    # the reference's detection constructor spells the argument `feedfoward_scale` (det tim.py:18-35)
    kw["feedfoward_scale" if getattr(cfg, "variant", "recognition") == "detection" else "feedforward_scale"] = cfg.feedforward_scale
    m = TIM(cfg.num_class, **kw)
    sd = synth.make_state_dict(cfg, seed=seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev), sd


def make_batch(cfg, B, nv, na, seed, dev):
    inp = synth.make_inputs(cfg, B, nv, na, seed=seed)
    return {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}


def make_det_targets(cfg, B, ngt, seed, dev):
    """synthetic ground truth of the detection windows in the wire format of det sliding_window.py:383-390: `ngt` sorted
    segments per window inside [0, 1] and their verb / noun / action (class_id: audio) labels"""
    rs = np.random.RandomState(seed)
    seg = lambda: torch.from_numpy(np.sort(rs.rand(B, ngt, 2), axis=-1).astype(np.float32)).to(dev)
    nc = cfg.num_class
    vc = nc[0] if isinstance(nc[0], (list, tuple)) else (nc[0], nc[0], nc[0])
    ri = lambda hi: torch.from_numpy(rs.randint(0, hi, (B, ngt))).to(dev)
    return {"v_gt_segments": seg(), "a_gt_segments": seg(), "verb": ri(vc[0]), "noun": ri(vc[1]), "action": ri(vc[2]),
            "class_id": ri(nc[1])}


def det_train_step(model, batch, target, state):
    """One TRUE detection training step (det scripts/train.py:212-349): model.train(), the query set drawn from the training
    pyramid, IoU labelling of every query on the device (timhip_label_queries), encoder forward, focal classification loss
    with IoU row weights + 1-D DIoU regression loss over the positive queries, full backward.  No optimizer (as the
    recognition step: the operand-copy refresh of the weights stands for it)."""
    from tim_amd import losses
    inner = model.module if hasattr(model, "module") else model
    for p in inner.parameters():
        p.grad = None
    inner.rt.invalidate_weights()
    output, offsets, labels, queries, ious = model([batch["visual"], batch["audio"]], "encoder", batch["times"], target, label_queries=True)
    loss = None
    sides = []
    if "visual" in inner.data_modality:
        ids = (0, 1, 2) if inner.include_verb_noun else (2,)
        sides.append((ids, 0, [labels[0][i] for i in ids]))
    if "audio" in inner.data_modality:
        sides.append(((3,), 1, [labels[1]]))
    for mi, (cls_ids, reg_id, lab) in enumerate(sides):
        # the side's loss in a handful of launches (tim_amd/losses.py:detection_side_loss - flags, weights, positive count, EMA
        # normaliser and the divisions inside the kernels; the reference's `max(num_pos, 1)` compares a device tensor with a
        # Python int, i.e. synchronises the host every step).  The normaliser is ONE persistent device scalar advanced in place:
        # a captured step (HIP-graph replay) then advances the average on every replay.  TIM_AMD_DET_LOSS=composed: the same
        # value from the separate focal / DIoU functions and torch glue (the round-4 form; tests compare the two)
        if ("norm", mi) not in state:
            state[("norm", mi)] = torch.full((), 250.0, dtype=torch.float32, device=ious[reg_id].device)      # parser.py:113-121
        iou, off = ious[reg_id], offsets[reg_id]
        if os.environ.get("TIM_AMD_DET_LOSS", "fused") != "composed":
            side = losses.detection_side_loss([output[0][c] for c in cls_ids], lab, output[1][reg_id], off, iou, state[("norm", mi)],
                                              inner.iou_threshold, lambda_reg=0.5, momentum=0.9)   # lambda_reg = 0.5 (parser.py:78)
        else:
            valid_reg = off[:, 0] != float("inf")                       # det train.py:223
            valid_cls = iou >= 0.0
            w = torch.where(iou < inner.iou_threshold, torch.ones_like(iou), iou)   # :228
            num_pos = valid_reg.sum()
            normaliser = 0.9 * state[("norm", mi)] + 0.1 * torch.clamp(num_pos, min=1).to(torch.float32)   # :230
            state[("norm", mi)].copy_(normaliser.detach())
            cls = sum(losses.focal_loss_sum(output[0][c], lab[j], row_weights=w, row_valid=valid_cls)
                      for j, c in enumerate(cls_ids)) / (len(cls_ids) * normaliser)
            reg = losses.diou_loss_sum(output[1][reg_id], torch.where(valid_reg[:, None], off, torch.zeros_like(off)),
                                       row_valid=valid_reg) * 0.5 / normaliser   # :277-285
            side = cls + reg
        loss = side if mi == 0 else loss + side
    loss.backward()
    return {"loss": loss.detach(), "output": output, "offsets": offsets, "labels": labels, "queries": queries, "ious": ious}


def make_rec_targets(cfg, B, nv, na, seed, dev):
    """synthetic labels of a recognition batch in the loader's wire format (sliding_window.py:383-390): verb / noun / action
    per visual query, class_id per audio query, -1 where a window has fewer queries than the batch maximum (every fifth / fourth
    slot here: train.py:223-224 masks them by target != -1)"""
    rs = np.random.RandomState(seed)
    vc, ac = cfg.num_class[0], cfg.num_class[1]
    t = {"verb": rs.randint(0, vc[0], (B, nv)), "noun": rs.randint(0, vc[1], (B, nv)), "action": rs.randint(0, vc[2], (B, nv)),
         "class_id": rs.randint(0, ac, (B, na))}
    pad_v = rs.rand(B, nv) < 0.2
    for k in ("verb", "noun", "action"):
        t[k][pad_v] = -1
    t["class_id"][rs.rand(B, na) < 0.25] = -1
    return {k: torch.from_numpy(v.astype(np.int64)).to(dev) for k, v in t.items()}


def rec_train_step(model, batch, nv, na, st):
    """One recognition TRAINING ITERATION as recognition/scripts/train.py:184-366 runs it: time MLP -> mixup of the inputs
    (utils/mixup.py:4-22: three lerps with a batch permutation) -> encoder -> label-smoothed mixup cross entropy on the four
    heads (train.py:218-316, one fused launch pair per head: tim_amd/losses.py) + cross-modal dense relative localisation loss
    (train.py:331-336, lambda 0.3, m = 32) -> backward -> clip_grad_norm_(1.0) (train.py:358) -> AdamW (train.py:66,362; fused,
    capturable so that the whole iteration replays as one HIP graph) -> operand-copy refresh of the weights at the head of the
    next step.  lam, the permutation and the DRLoc positions are drawn once per process (a captured step freezes host-side
    draws anyway; their values do not change the cost of the step).  No GradScaler: the fp16 mode scales its gradient
    operands on the device (timhip_grad_scale) and the parameter gradients are true-scale fp32."""
    from tim_amd import losses
    inner = model.module if hasattr(model, "module") else model
    opt = st["opt"]
    opt.zero_grad(set_to_none=True)
    inner.rt.invalidate_weights()          # the optimizer changed the weights: redo the operand copies (one grouped launch)
    lam, perm, tgt, tgt_b = st["lam"], st["perm"], st["target"], st["target_b"]
    te = model(batch["times"], "time_mlp")
    vis, aud, te = [torch.lerp(t[perm], t, lam) for t in (batch["visual"], batch["audio"], te)]   # lam * t + (1 - lam) * t[perm]
    (verb, noun, action, audio), feats = model([vis, aud], "encoder", te, nv, na)
    ce = lambda x, k: losses.mixup_cross_entropy(x, tgt[k].reshape(-1), tgt_b[k].reshape(-1), lam, 0.2)   # noqa: E731
    loss = (ce(verb, "verb") + ce(noun, "noun") + ce(action, "action")) / 3.0 + 1.0 * ce(audio, "class_id")
    nf = inner.cfg.num_feats
    loss = loss + 0.3 * losses.dense_relative_localization_loss_crossmodal(feats[:, :nf], feats[:, nf:], inner, 32, positions=st["pos"])
    loss.backward()
    torch.nn.utils.clip_grad_norm_(st["params"], 1.0, foreach=True)
    opt.step()
    return loss.detach()


def make_rec_train_state(model, cfg, B, nv, na, dev, seed=0):
    inner = model.module if hasattr(model, "module") else model
    params = list(inner.parameters())
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(B, generator=g).to(dev)
    tgt = make_rec_targets(cfg, B, nv, na, seed + 7, dev)
    nf = cfg.num_feats
    return {"rec": True, "params": params, "lam": float(np.random.RandomState(seed).beta(0.2, 0.2)) * 0.5 + 0.25, "perm": perm,
            "target": tgt, "target_b": {k: v[perm] for k, v in tgt.items()},
            "pos": (torch.randint(nf, (B, 32), generator=g).to(dev), torch.randint(nf, (B, 32), generator=g).to(dev)),
            # lr 1e-4 / weight decay 1e-4: parser.py:121-131; capturable: the step counter lives on the device
            "opt": torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, fused=True, capturable=True)}


def step_fn(model, batch, nv, na, R):
    inner = model.module if hasattr(model, "module") else model
    if isinstance(R, dict) and R.get("rec"):     # recognition as the reference's training iteration (losses, clip, AdamW)
        return rec_train_step(model, batch, nv, na, R)
    if isinstance(R, dict):          # detection as true training: R carries {"target": ..., state}
        return det_train_step(model, batch, R["target"], R)
    plist = inner.__dict__.get("_bench_plist")
    if plist is None:
        plist = inner.__dict__["_bench_plist"] = list(inner.parameters())   # (walking the module tree costs ~0.25 ms per step)
    for p in plist:
        p.grad = None
    inner.rt.invalidate_weights()  # weights changed (optimizer step): redo the operand copies
    if hasattr(inner, "reg_head"):   # detection in inference form with gradients: dense query pyramid generated by the model
        (cls, reg, feats), _, _, _, _ = model([batch["visual"], batch["audio"]], "encoder", batch["times"], None,
                                              label_queries=False)
        outs = [t for t in list(cls) + list(reg) + [feats] if t is not None]
        if R[0] is None:
            g = torch.Generator(device="cpu").manual_seed(1)
            R[:] = [torch.randn(o.shape, generator=g).to(o.device) * 0.05 for o in outs]
        torch.autograd.backward(outs, R)
        return
    te = model(batch["times"], "time_mlp")
    (verb, noun, action, audio), feats = model([batch["visual"], batch["audio"]], "encoder", te, nv, na)
    outs = [t for t in (verb, noun, action, audio, feats) if t is not None]
    if R[0] is None:
        g = torch.Generator(device="cpu").manual_seed(1)
        R[:] = [torch.randn(o.shape, generator=g).to(o.device) * 0.05 for o in outs]
    torch.autograd.backward(outs, R)


def gemm_roofline(model, cfg, B, nv, na, precision, reps=20):
    """Per-launch timing of the dominant kernel family (the MFMA NT GEMM) with HIP events on the launch
    stream, for every GEMM shape of one encoder layer; algorithmic FLOPs = 2*M*N*K per launch."""
    from tim_amd import _lib as L
    rt = model.rt
    dev = next(model.parameters()).device
    E, FF = cfg.E, cfg.FF
    S = cfg.F + cfg.num_queries(nv, na)
    M = B * S
    Mp = (M + 63) // 64 * 64
    shapes = [  # name, M, N, K, epi, launches per layer (fwd + bwd)
        ("in_proj fwd", M, 3 * E, E, L.EPI_STORE_T, 1), ("out_proj fwd", M, E, E, L.EPI_DROP_RES_F32, 1),
        ("ffn1 fwd", M, FF, E, L.EPI_GELU_DROP_G2, 1), ("ffn2 fwd", M, E, FF, L.EPI_DROP_RES_F32, 1),
        # (round 3: the two N = E input-gradient products are stored 16-bit; LayerNorm-backward adds them to the fp32 stream)
        ("ffn2 dgrad", M, FF, E, L.EPI_MULAUX_T, 1), ("ffn1 dgrad", M, E, FF, L.EPI_STORE_T, 1),
        ("out_proj dgrad", M, E, E, L.EPI_STORE_T, 1), ("in_proj dgrad", M, E, 3 * E, L.EPI_STORE_T, 1),
    ]
    if precision in ("bf16", "fp16"):
        shapes.append(("layer wgrad (ffn2 + ffn1 + out_proj + in_proj, one grouped launch)", 0, 0, Mp, "group", 1))
    else:
        shapes += [("in_proj wgrad", 3 * E, E, Mp, "slab", 1), ("out_proj wgrad", E, E, Mp, "slab", 1),
                   ("ffn1 wgrad", FF, E, Mp, "slab", 1), ("ffn2 wgrad", E, FF, Mp, "slab", 1)]
    g = torch.Generator().manual_seed(3)
    out = []
    tot_flop = tot_ms = 0.0
    for name, m_, n_, k_, epi, cnt in shapes:
        if epi == "group":
            # the four weight gradients of one layer through timhip_wgrad_group, as timhip_layer_bwd_weights launches them
            items, fl = [], 0.0
            for (no, ko) in ((E, FF), (FF, E), (E, E), (3 * E, E)):
                dY = torch.randn(M, no, generator=g).to(dev).to(rt.op_dtype)
                Xa = torch.randn(M, ko, generator=g).to(dev).to(rt.op_dtype)
                items.append((dY, no, Xa, ko, torch.zeros((no, ko), dtype=torch.float32, device=dev),
                              torch.zeros(no, dtype=torch.float32, device=dev)))
                fl += 2.0 * no * ko * M
            for _ in range(3):
                rt.wgrad_group(items, M, accumulate=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                rt.wgrad_group(items, M, accumulate=False)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            out.append({"gemm": name, "M": 12 * E * E // max(E, 1), "N": E, "K": M, "us": round(ms * 1e3, 2),
                        "tflops": round(fl / ms / 1e9, 1)})
            tot_flop += fl * cnt
            tot_ms += ms * cnt
            continue
        # the fp16 mode carries the out-projection weight as [w_hi | w_lo]: K' = 2K, the A operand read twice (TimEpi.a_wrap_k);
        # algorithmic FLOPs stay 2 M N K
        wrap = k_ if (name == "out_proj fwd" and getattr(rt, "split_outproj", False)) else 0
        A = (torch.randn(m_, k_, generator=g)).to(dev).to(rt.op_dtype)
        Bm = (torch.randn(n_, k_ * (3 if wrap else 1), generator=g) * k_ ** -0.5).to(dev).to(rt.op_dtype)
        o0 = torch.zeros((m_, n_), dtype=torch.float32, device=dev)
        o1 = torch.zeros((m_, n_), dtype=torch.float32, device=dev)
        res = torch.zeros((m_, n_), dtype=torch.float32, device=dev)
        bias = torch.zeros(n_, device=dev)
        sk = 1
        if epi == "slab":
            # weight gradient dW[m_, n_] += dY[M, m_]^T X[M, n_] through timhip_wgrad (transposing-read TN
            # MFMA kernel + slab reduce), timed as one unit; M = B*S rows
            dY = torch.randn(M, m_, generator=g).to(dev).to(rt.op_dtype)
            Xa = torch.randn(M, n_, generator=g).to(dev).to(rt.op_dtype)
            dW = torch.zeros((m_, n_), dtype=torch.float32, device=dev)
            dbv = torch.zeros(m_, dtype=torch.float32, device=dev)
            for _ in range(3):
                rt.wgrad(dY, m_, Xa, n_, M, dW, dbv)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                rt.wgrad(dY, m_, Xa, n_, M, dW, dbv)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fl = 2.0 * m_ * n_ * M
            out.append({"gemm": name, "M": m_, "N": n_, "K": M, "us": round(ms * 1e3, 2),
                        "tflops": round(fl / ms / 1e9, 1)})
            tot_flop += fl * cnt
            tot_ms += ms * cnt
            continue
        else:
            kw = dict(out1=o1, ld1=n_, bias=None if epi in (L.EPI_ADD_F32, L.EPI_DGELU_T, L.EPI_MULAUX_T) else bias,
                      res=res, ldres=n_, aux=o1, ldaux=n_, p_drop=cfg.enc_dropout, seed=7, site=5, splitk=sk)
            if wrap:
                kw.update(a_wrap_k=wrap, rep=2)
        kk = 2 * k_ if wrap else k_
        for _ in range(3):
            rt.gemm(epi, A, Bm, m_, n_, kk, o0, n_, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            rt.gemm(epi, A, Bm, m_, n_, kk, o0, n_, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * m_ * n_ * k_
        out.append({"gemm": name + (" (weight as hi + lo halves: K' = 2K)" if wrap else ""), "M": m_, "N": n_, "K": k_,
                    "us": round(ms * 1e3, 2), "tflops": round(fl / ms / 1e9, 1)})
        tot_flop += fl * cnt
        tot_ms += ms * cnt
    return out, tot_flop, tot_ms


def _oracle_masks(cfg, B, S, seed=0):
    """Bernoulli keep-masks for every dropout site of one training forward, in the layout oracle/tim_oracle.py takes them
    (which bits are kept does not change the cost of the step)"""
    g = torch.Generator().manual_seed(seed)
    keep = lambda p, *shape: (torch.rand(*shape, generator=g) >= p).to(torch.uint8)
    nf, E, FF, H, F = cfg.num_feats, cfg.E, cfg.FF, cfg.nhead, cfg.F
    m = {"feat_visual": keep(cfg.feat_drop, B, nf, cfg.visual_input_dim), "feat_audio": keep(cfg.feat_drop, B, nf, cfg.audio_input_dim),
         "seq": keep(cfg.seq_drop, B, S, E)}
    for l in range(cfg.num_layers):
        m["l%d_attn" % l] = keep(cfg.enc_dropout, B, H, S, F + 1)
        m["l%d_drop1" % l] = keep(cfg.enc_dropout, B, S, E)
        m["l%d_ffn" % l] = keep(cfg.enc_dropout, B, S, FF)
        m["l%d_drop2" % l] = keep(cfg.enc_dropout, B, S, E)
    return m


def _oracle_rate(cfg, sd_np, nv, na, threads, B, runs=3):
    """interval-queries/s of the CPU oracle on THE STEP THE GPU LEG TIMES: training-mode forward (all five dropout sites, masks
    supplied) + backward to every parameter gradient of B windows; one untimed step, then `runs` timed ones -> median"""
    from oracle import tim_oracle as O
    torch.set_num_threads(threads)
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in sd_np.items()}
    inp = {k: torch.from_numpy(v) for k, v in synth.make_inputs(cfg, B, nv, na, seed=11).items()}
    masks = _oracle_masks(cfg, B, cfg.F + cfg.num_queries(nv, na))

    def one():
        for v in sd.values():
            v.grad = None
        cls, feats = O.forward(sd, cfg, inp["visual"], inp["audio"], inp["times"], nv, na, masks=masks)
        loss = sum(c.sum() for c in cls if c is not None) + feats.sum()
        loss.backward()

    one()
    times = []
    for _ in range(runs):
        t0 = time.time()
        one()
        times.append(time.time() - t0)
    med = sorted(times)[len(times) // 2]
    return B * (nv + na) / med, times


def cpu_baseline(cfg, sd_np, nv, na, workload, B=64):
    """The oracle (torch CPU fp32 restatement of the reference path, kind "port") timed on this box's host cores on the same
    step the GPU leg times - train-mode forward + backward of B = 64 windows of the same shapes - at the thread count where
    torch's CPU GEMMs of this size stop scaling (<= 32): one untimed step, median of three timed ones."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))
    if avail < 16:
        B = min(B, 4)     # (a small host, e.g. the build container: keep the leg bounded)
    v, times = _oracle_rate(cfg, sd_np, nv, na, cores, B)
    return {"value": round(v, 2), "unit": "interval-queries/s", "cores": cores, "kind": "port",
            "host_threads_total": os.cpu_count(), "host_threads_available": avail,
            "math": "train-mode forward (feature / sequence / attention / residual / FFN dropout, masks supplied) + backward to every "
                    "parameter gradient: the step the GPU leg times",
            "sample": "oracle/tim_oracle.py fwd+bwd of B=%d windows of the %s shapes (%d interval queries per step), torch CPU fp32, %d "
                      "threads: 1 untimed step, median of %d timed steps (%s s)"
                      % (B, workload, B * (nv + na), cores, len(times), ", ".join("%.2f" % t for t in times))}


def logit_parity(model, cfg, sd_np, nv, na, dev, B=4):
    """max |logit - fp32 CPU oracle| of `model` (eval mode) on B synthetic windows of the bench shapes - measured in this run"""
    from oracle import tim_oracle as O
    inp_np = synth.make_inputs(cfg, B, nv, na, seed=12)
    was = model.training
    model.eval()
    with torch.no_grad():
        d = {k: torch.from_numpy(v).to(dev) for k, v in inp_np.items()}
        te = model(d["times"], "time_mlp")
        heads, _ = model([d["visual"], d["audio"]], "encoder", te, nv, na)
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
        ref, _ = O.forward(sd, cfg, torch.from_numpy(inp_np["visual"]), torch.from_numpy(inp_np["audio"]),
                           torch.from_numpy(inp_np["times"]), nv, na)
    model.train(was)
    worst, n = 0.0, 0
    for a, b in zip(heads, ref):
        if a is not None:
            worst = max(worst, (a.float().cpu() - b).abs().max().item())
            n += a.numel()
    return worst, n


def det_logit_parity(model, cfg, sd_np, dev, B=2):
    """detection model (eval: the 399 inference queries, det tim.py:348) against the fp32 CPU oracle on B synthetic windows"""
    from oracle import tim_oracle as O
    inp_np = synth.make_inputs(cfg, B, 0, 0, seed=12)
    was = model.training
    model.eval()
    with torch.no_grad():
        d = {k: torch.from_numpy(v).to(dev) for k, v in inp_np.items()}
        (cls, reg, _), _, _, _, _ = model([d["visual"], d["audio"]], "encoder", d["times"], None, label_queries=False)
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
        q = O.generate_queries(0.01)
        times = torch.cat([torch.from_numpy(inp_np["times"]), q.expand(B, -1, -1)], 1)
        nq = q.shape[1]
        has_v = "visual" in cfg.data_modality
        rc, _, rr = O.forward(sd, cfg, torch.from_numpy(inp_np["visual"]), torch.from_numpy(inp_np["audio"]), times,
                              nq if has_v else 0, 0 if cfg.data_modality == "visual" else nq)
    model.train(was)
    worst, n = 0.0, 0
    for a, b in list(zip(cls, rc)) + list(zip(reg, rr)):
        if a is not None and b is not None:
            worst = max(worst, (a.float().cpu() - b).abs().max().item())
            n += a.numel()
    return worst, n


def secondary_block(workload, B, precision, dev, steps, warmup, det_train=False, graph=False, rec_train=False):
    """one more BASELINE.json configuration, same build and mode, measured like the headline (eager steps of the full
    forward + backward on HBM-resident synthetic inputs) plus its logit error against the fp32 oracle"""
    cfg = named_config(workload)
    det = getattr(cfg, "variant", "recognition") == "detection"
    nv, na = (10, 0) if workload == "C1" else (15, 10)
    if det:
        nv, na = 399, 0
    m, sd_np = build_model(cfg, precision, dev, seed=0)
    m.train(det_train or not det)
    batch = make_batch(cfg, B, 0 if det else nv, na, seed=100, dev=dev)
    R = {"target": make_det_targets(cfg, B, 6, 5, dev)} if det_train else (make_rec_train_state(m, cfg, B, nv, na, dev) if rec_train else [None])
    err_first = (logit_parity(m, cfg, sd_np, nv, na, dev, B=2) if rec_train else None)   # (before AdamW moves the weights)
    ms, ms_mean, ms_max = robust_step_ms(lambda: step_fn(m, batch, nv, na, R), steps, max(warmup, 10 if not rec_train else 30))
    host_issue = None
    if rec_train or B < 16:   # host time to ISSUE a step (what bounds the eager loop at small batches / long launch lists)
        torch.cuda.synchronize()
        th0 = time.perf_counter()
        for _ in range(5):
            step_fn(m, batch, nv, na, R)
        host_issue = (time.perf_counter() - th0) / 5 * 1e3
        torch.cuda.synchronize()
    err, nlog = err_first if err_first is not None else (det_logit_parity(m, cfg, sd_np, dev) if det else logit_parity(m, cfg, sd_np, nv, na, dev, B=2))
    qps = B * (nv + na) / ms * 1e3
    out = {"windows_per_gpu": B, "tokens_per_window": cfg.F + cfg.num_queries(nv, na), "ms_per_step": round(ms, 3),
           "interval_queries_per_s": round(qps, 1),
           "frac_of_mfma_peak": round(qps * GFLOP_PER_QUERY[workload] / 1e3 / PEAK_BF16_TFLOPS, 4),
           "max_abs_logit_err": float("%.3g" % err), "parity_sample": "%d outputs of 2 windows, eval mode, vs the fp32 CPU oracle" % nlog,
           "timing": "median of %d per-step HIP-event intervals after %d warm-ups (mean %.3f, max %.3f ms)"
                     % (steps, max(warmup, 10), ms_mean, ms_max)}
    if host_issue is not None:
        out["host_issue_ms_per_step"] = round(host_issue, 3)
    if graph:   # the same step as one HIP-graph replay (child process, as for the headline): the host-bound small model's fast path
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--graph-child", "--workload", workload, "--batch", str(B),
               "--precision", precision, "--steps", str(max(steps, 20)), "--warmup", str(warmup)] + (["--det-train"] if det_train else []) \
            + (["--rec-train"] if rec_train else [])
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out["graph_replay"] = json.loads(line[-1]) if line else {"error": "child exited with %d" % r.returncode}
            g = out["graph_replay"].get("ms_per_step")
            if g:
                out["graph_replay"]["frac_of_mfma_peak"] = round(B * (nv + na) / g * GFLOP_PER_QUERY[workload] / PEAK_BF16_TFLOPS, 4)
        except Exception as e:  # noqa: BLE001
            out["graph_replay"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    del m, batch
    return out


def timing_stop_families():
    """-> [(ms, work, launches)] for the GEMM (FLOPs), attention and LayerNorm (algorithmic bytes) launches bracketed since
    timhip_gemm_timing_start (tim_amd/csrc/timing.hip)"""
    import ctypes as C
    from tim_amd import _lib as L
    ms, wk, nl = (C.c_double * 3)(), (C.c_double * 3)(), (C.c_int * 3)()
    L.call("timhip_timing_stop_families", C.cast(ms, C.c_void_p), C.cast(wk, C.c_void_p), C.cast(nl, C.c_void_p))
    return [(ms[i], wk[i], nl[i]) for i in range(3)]


def robust_step_ms(fn, steps, warmup):
    """-> (median, mean, max) milliseconds per call of fn(): one HIP event in front of every step and one behind the last, on the
    stream the step is issued on, read after one synchronisation.  The secondary blocks used to report one wall-clock mean
    over 10 steps after 3 warm-ups: a single allocator growth or module load inside those 10 steps owned the figure (round 3:
    the driver's box showed 15.9 ms for a step that runs in 5.9) - the median of per-step intervals does not care."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for i in range(steps):
        evs[i].record()
        fn()
    evs[steps].record()
    torch.cuda.synchronize()
    d = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    return d[len(d) // 2], sum(d) / len(d), d[-1]


def timed_steps(model, batch, nv, na, steps, warmup):
    R = [None]
    for _ in range(warmup):
        step_fn(model, batch, nv, na, R)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn(model, batch, nv, na, R)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--batch", type=int, default=64, help="windows per GPU")
    ap.add_argument("--workload", default="C2a")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra-step", action="store_true",
                    help="skip the untimed extras - the step in the other stream configuration and the forward-only loop "
                         "(profiling runs: the summary then holds only steps like the timed ones)")
    ap.add_argument("--no-graph", action="store_true", help="skip the HIP-graph replay measurement (implies --step-mode eager)")
    ap.add_argument("--step-mode", default="auto", choices=["auto", "graph", "eager"],
                    help="how the K timed steps are issued: 'graph' = the whole step (weight refresh, forward, backward) captured once "
                         "in a HIP graph and replayed K times (tim_amd/graph.py: one host call per step); 'eager' = K eager steps of "
                         "~144 launches each; 'auto' (default) = graph on one GPU, eager when the step contains collectives (--gpus > 1)")
    ap.add_argument("--graph-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--det-train", action="store_true",
                    help="detection workloads (C4): the true training step (train-mode query draw, labelling, focal + DIoU) instead of the inference form")
    ap.add_argument("--rec-train", action="store_true",
                    help="recognition workloads: the reference's whole training iteration (mixup, four mixup cross entropies + cross-modal "
                         "DRLoc, clip_grad_norm_, AdamW) instead of the fixed-cotangent forward + backward the metric is defined on")
    ap.add_argument("--no-repeat", action="store_true", help="skip the four repeat sets of K steps behind the timed region")
    ap.add_argument("--no-per-shape", action="store_true", help="skip the isolated per-shape GEMM loop (profiling runs)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary blocks (parity sample, bf16 mode, C2b)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py: --gpus %d must be launched with torch.distributed.run (WORLD_SIZE unset)" % args.gpus,
              file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    # TIM_AMD_BENCH_SHARE_GPU=1 (tests/test_gpu_bench_ranks.py): run the multi-rank control flow on a one-GPU box - all
    # ranks on GPU 0, collectives over gloo (RCCL refuses two ranks on one device).  Not a measurement mode.
    share = os.environ.get("TIM_AMD_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # TIM_AMD_BENCH_FORCE_DP=1 (tests/test_gpu_bench_ranks.py): the data-parallel code path of this file on ONE rank - a one-rank
    # RCCL group with the exchange forced on (every collective a copy).  Not a measurement mode either.
    force_dp = world == 1 and os.environ.get("TIM_AMD_BENCH_FORCE_DP", "0") == "1"
    dp_on = world > 1 or force_dp
    if dp_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dp:
            os.environ.setdefault("MASTER_PORT", "29517")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = named_config(args.workload)
    nv, na = (10, 0) if args.workload == "C1" else (15, 10)
    detection = getattr(cfg, "variant", "recognition") == "detection"
    if detection:
        nv, na = 399, 0          # the dense query pyramid the detection model generates (det tim.py:140-155)
    B = args.batch
    model, sd_np = build_model(cfg, args.precision, dev, seed=0)
    model.train(not detection or args.det_train)   # C4 without --det-train: inference form with gradients (no GT targets needed)
    run_model = model
    if dp_on:
        from tim_amd.dp import DataParallel
        # fp32 on the wire = the exact mean the reference's DistributedDataParallel computes (the default of tim_amd.dp);
        # TIM_AMD_DP_WIRE=bf16 selects the half-traffic exchange (bf16 payload, fp32 accumulation, all-to-all form: eager only)
        wire = torch.bfloat16 if os.environ.get("TIM_AMD_DP_WIRE", "fp32") == "bf16" else torch.float32
        run_model = DataParallel(model, wire_dtype=wire, force=force_dp, buckets_per_exchange=int(os.environ.get("TIM_AMD_DP_BUCKETS", "4")))
    batch = make_batch(cfg, B, 0 if detection else nv, na, seed=100 + rank, dev=dev)  # each rank its own shard of windows
    R = {"target": make_det_targets(cfg, B, 6, 5 + rank, dev)} if (detection and args.det_train) else [None]
    if args.rec_train and not detection:
        R = make_rec_train_state(model, cfg, B, nv, na, dev, seed=rank)

    def barrier():
        if dp_on:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    if args.graph_child:
        from tim_amd.graph import GraphedStep
        res = {}
        try:
            gstep = GraphedStep(model, lambda: step_fn(model, batch, nv, na, R))
            for _ in range(args.warmup):
                gstep()
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            for _ in range(args.steps):
                gstep()
            torch.cuda.synchronize()
            graph_ms = (time.perf_counter() - tg0) / args.steps * 1e3
            res = {"ms_per_step": round(graph_ms, 3), "interval_queries_per_s": round(B * (nv + na) / graph_ms * 1e3, 1),
                   "steps": args.steps}
        except Exception as e:  # noqa: BLE001
            res = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        print(json.dumps(res), flush=True)
        os._exit(0)   # skip interpreter teardown of a process that may hold a half-built capture
    # ---- how the timed steps are issued.  The eager step is ~144 launches that a free host issues in ~3.4 ms - under the 5.3 ms
    # the GPU needs, but not by much: on a box whose host cores were busy with other tenants (load average 10-24, round 4) the
    # same build measured 9.25 ms per eager step with 8.2 ms of it host issue, and 5.27 ms as a replayed HIP graph.  The headline
    # therefore times the step the way the framework runs it in production (tim_amd/graph.py): captured once,
    # replayed - every kernel of the eager step, fresh dropout masks per replay (device-side salt), the weight refresh included;
    # parity of a replay with the eager step: tests/test_gpu_graph.py (C2a B = 64 fp16, detection training).  The eager figures
    # stay in the line (`eager`), and with collectives in the step (--gpus > 1) the timed steps are eager.
    mode = args.step_mode
    if args.no_graph and mode == "auto":
        mode = "eager"
    # The data-parallel step is captured too when its exchange is capturable: reduce-scatter + all-gather (or all-reduce) over RCCL
    # (tim_amd/dp.py "rs_ag": the default for the fp32 wire; all_to_all_single's send / receive pairs hang or crash
    # hipStreamEndCapture on this stack - profiles/r05_rccl_capture_probe.txt).  A capture is LOCAL (collectives are recorded, not
    # run), so a rank whose capture fails cannot strand its peers; the ranks then agree (MIN over a flag) whether EVERYBODY
    # replays or everybody issues eagerly.  TIM_AMD_BENCH_DP_GRAPH=0: eager data-parallel steps.
    dp_capturable = dp_on and not share and getattr(run_model, "collective", "a2a") in ("rs_ag", "allreduce") \
        and os.environ.get("TIM_AMD_BENCH_DP_GRAPH", "1") != "0"
    if mode == "auto":
        mode = "graph" if (not dp_on or dp_capturable) else "eager"
    if mode == "graph" and dp_on and not dp_capturable:
        raise SystemExit("--step-mode graph: this data-parallel exchange cannot be captured (collective %r over %s)"
                         % (getattr(run_model, "collective", None), "gloo" if share else "RCCL"))
    eager = None
    live = None
    fam = None
    if mode == "graph":
        # the same step issued eagerly, right BEFORE the capture and the timed region: its wall-clock rate and host issue time (the
        # robustness margin the replay buys), and - in its last step - the per-launch HIP events of the roofline: a replayed graph
        # cannot carry events around individual launches; the kernels, their launch parameters and the stream are the eager
        # step's.  (Before, not behind: eager steps issued in a process that holds an instantiated graph of the same step cost
        # the host 4 - 11 ms instead of 3.5, tools/eager_after_graph.py and profiles/r04_c_bench_default.json.)
        ne = max(5, min(args.steps, 20))
        for _ in range(max(30, args.warmup)):   # (a process needs ~30 eager steps before its host side is in steady state: lazily
            step_fn(run_model, batch, nv, na, R)   # loaded code objects, allocator growth - tools/eager_after_graph.py: 6.1 -> 3.5 ms of issue time)
        torch.cuda.synchronize()
        te0 = time.perf_counter()
        for i in range(ne):
            if i == ne - 1 and rank == 0 and not args.no_roofline:
                from tim_amd import _lib as L
                L.call("timhip_gemm_timing_start", 512, 1.0e10)
                live = True
            step_fn(run_model, batch, nv, na, R)
        te_issue = time.perf_counter() - te0
        torch.cuda.synchronize()
        eager = {"ms_per_step": round((time.perf_counter() - te0) / ne * 1e3, 3), "host_issue_ms_per_step": round(te_issue / ne * 1e3, 3),
                 "steps": ne}
        if live:
            fam = timing_stop_families()
            live = fam[0]
    gstep = None
    graph_note = None
    if mode == "graph":
        try:
            from tim_amd.graph import GraphedStep
            gstep = GraphedStep(run_model, lambda: step_fn(run_model, batch, nv, na, R), count_nodes=True)
        except Exception as e:  # noqa: BLE001  (a runtime that refuses the capture: the eager steps are timed, and the line says so)
            gstep, mode = None, "eager"
            graph_note = "HIP-graph capture failed (%s: %s): eager steps timed" % (type(e).__name__, str(e)[:160])
        if dp_on:   # everybody replays, or nobody does
            import torch.distributed as dist
            okf = torch.tensor([1.0 if gstep is not None else 0.0], device=dev)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if okf.item() < 0.5 and gstep is not None:
                gstep.reset()
                gstep, mode = None, "eager"
                graph_note = "HIP-graph capture failed on another rank: eager steps timed"
    run_step = gstep if gstep is not None else (lambda: step_fn(run_model, batch, nv, na, R))
    for _ in range(args.warmup):
        run_step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if gstep is None and live is None and i == args.steps - 1 and rank == 0 and not args.no_roofline:
            # roofline: HIP events around every encoder-layer GEMM launch (>= 1e10 FLOPs: excludes heads / embedders) of the
            # LAST timed step, recorded on the streams the kernels are launched on (timhip_gemm_timing_*)
            from tim_amd import _lib as L
            L.call("timhip_gemm_timing_start", 512, 1.0e10)
            live = True
        run_step()
    t_enqueue = time.perf_counter() - t0   # host time to ISSUE the K steps (the GPU may still be running them)
    barrier()
    dt = time.perf_counter() - t0
    if dp_on:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    import ctypes as C
    from tim_amd import _lib as L
    ms_, fl_, n_ = C.c_double(0), C.c_double(0), C.c_int(0)
    if live is True:
        fam = timing_stop_families()
        live = fam[0]
    # the same K steps four more times (same barriers, same step): the line carries its own spread.  `value` / `ms_per_step`
    # stay the FIRST set - the contract's timed region; `repeat_ms` lists all five, `median_ms` their median
    repeat_ms = [dt / args.steps * 1e3]
    for _ in range(0 if args.no_repeat else 4):
        barrier()
        tr0 = time.perf_counter()
        for _i in range(args.steps):
            run_step()
        barrier()
        dtr = time.perf_counter() - tr0
        if dp_on:
            import torch.distributed as dist
            t = torch.tensor([dtr], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtr = t.item()
        repeat_ms.append(dtr / args.steps * 1e3)
    comm = None
    if dp_on:
        # what the gradient exchange costs the step: (a) the same steps with the exchange switched off (every rank, same
        # barriers; replays of a second captured graph when the timed steps were replays) -> exposed time = difference;
        # (b) one eager step with events on the comm stream -> its busy time and bus rate
        import torch.distributed as dist
        nx = max(3, min(10, args.steps))
        with run_model.no_sync():
            ns_step = lambda: step_fn(run_model, batch, nv, na, R)   # noqa: E731
            if gstep is not None:
                try:
                    from tim_amd.graph import GraphedStep
                    ns_step = GraphedStep(run_model, lambda: step_fn(run_model, batch, nv, na, R))
                except Exception:  # noqa: BLE001
                    pass
                okf = torch.tensor([1.0 if not callable(getattr(ns_step, "reset", None)) else 2.0], device=dev)
                dist.all_reduce(okf, op=dist.ReduceOp.MIN)
                if okf.item() < 1.5 and callable(getattr(ns_step, "reset", None)):   # (somebody's capture failed: eager everywhere)
                    ns_step.reset()
                    ns_step = lambda: step_fn(run_model, batch, nv, na, R)   # noqa: E731
            ns_step()
            barrier()
            tx0 = time.perf_counter()
            for _ in range(nx):
                ns_step()
            barrier()
            t = torch.tensor([time.perf_counter() - tx0], device=dev, dtype=torch.float64)
            ns_graph = callable(getattr(ns_step, "reset", None))
            if ns_graph:
                ns_step.reset()
            del ns_step
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_nosync = t.item() / nx * 1e3
        if not share:
            run_model.begin_step_timing()
            step_fn(run_model, batch, nv, na, R)
            comm_ms, wire_bytes = run_model.end_step_timing()
        else:
            comm_ms, wire_bytes = None, None
        barrier()
        forms = {"rs_ag": "reduce-scatter + all-gather per range of gradient buckets (RCCL sums in fp32 on the way)",
                 "a2a": "all-to-all + local fp32 sum + all-gather per range of gradient buckets",
                 "allreduce": "one fp32 all-reduce per range of gradient buckets"}
        comm = {"collective": run_model.collective,
                "ms_per_step_without_exchange": round(ms_nosync, 3),
                "without_exchange_timed_as": "hip_graph_replay" if ns_graph else "eager",
                "exposed_comm_ms": round(dt / args.steps * 1e3 - ms_nosync, 3),
                "comm_stream_busy_ms": None if comm_ms is None else round(comm_ms, 3),
                "bytes_sent_plus_received_per_rank": wire_bytes,
                "bus_GBps_per_rank": None if not comm_ms else round(wire_bytes / comm_ms / 1e6, 1),
                "wire_dtype": str(run_model.wire_dtype).replace("torch.", ""),
                "note": forms.get(run_model.collective, run_model.collective) + ", on a side stream under the backward "
                        "(tim_amd/dp.py); fp32 payload by default (the exact mean, collectives run on the bucket itself), "
                        "TIM_AMD_DP_WIRE=bf16 halves the traffic (all-to-all form); xGMI peak 7 links x ~153 GB/s per GPU"}
    live_serial = None
    if live and not dp_on and not args.no_extra_step:
        # the same measurement on one extra (untimed) step in the other stream configuration (weight gradients on a side
        # stream when the timed steps ran single-stream, and vice versa).  Single-GPU runs only: a step of the data-parallel
        # model contains collectives, and the other ranks are past their last step.
        was = model.rt.overlap_wgrad
        model.rt.overlap_wgrad = not was
        L.call("timhip_gemm_timing_start", 256, 1.0e10)
        step_fn(run_model, batch, nv, na, R)
        torch.cuda.synchronize()
        L.call("timhip_gemm_timing_stop", C.byref(ms_), C.byref(fl_), C.byref(n_))
        model.rt.overlap_wgrad = was
        live_serial = fl_.value / ms_.value / 1e9 if ms_.value > 0 else None

    queries_per_step = world * B * (nv + na)
    value = queries_per_step * args.steps / dt
    # forward-only rate (SURVEY 8d asks for it beside the fwd+bwd metric): same batch, train-mode dropout, no autograd graph
    fwd_ms = None
    if rank == 0 and not dp_on and not detection and not args.no_extra_step:
        with torch.no_grad():
            for _ in range(3):
                te_ = model(batch["times"], "time_mlp")
                model([batch["visual"], batch["audio"]], "encoder", te_, nv, na)
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            for _ in range(10):
                te_ = model(batch["times"], "time_mlp")
                model([batch["visual"], batch["audio"]], "encoder", te_, nv, na)
            torch.cuda.synchronize()
            fwd_ms = (time.perf_counter() - tf0) / 10 * 1e3
    gq = GFLOP_PER_QUERY.get(args.workload)
    out = {
        "metric": "interval-queries/sec (fwd+bwd), d=512 L=6 EPIC-100 window",
        "value": round(value, 1), "unit": "interval-queries/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "repeat_ms": [round(x, 3) for x in repeat_ms], "median_ms": round(sorted(repeat_ms)[len(repeat_ms) // 2], 3),
        "host_issue_ms_per_step": round(t_enqueue / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": {"bf16": "bf16", "fp16": "fp16"}.get(args.precision, "f32"),
        "data": "synthetic",
        "step_mode": ("hip_graph_replay: the whole step (weight refresh, forward, backward) captured once by tim_amd.graph.GraphedStep, "
                      "K replays timed" if gstep is not None else "eager: K eager steps timed") + ("; " + graph_note if graph_note else ""),
        "config": {"workload": "%s: EPIC-100 A+V recognition, d_model 512 (E 1024), 6 layers, 8 heads, 50+50 "
                               "feature tokens, 15+10 interval queries, train-mode dropout" % args.workload
                   if args.workload == "C2a" else args.workload,
                   "windows_per_gpu": B, "tokens_per_window": cfg.F + cfg.num_queries(nv, na),
                   "global_batch": world * B, "parallelism": "dp%d" % world, "precision": args.precision},
    }
    out["windows_per_s"] = round(world * B * args.steps / dt, 1)
    try:   # which box, and how busy its host was (the eager figures depend on it; the replay should not)
        import socket
        out["box"] = {"hostname": socket.gethostname(), "loadavg_1_5_15": [round(x, 2) for x in os.getloadavg()],
                      "host_threads": os.cpu_count(), "gpu": torch.cuda.get_device_name(dev)}
    except Exception:  # noqa: BLE001
        pass
    if force_dp:
        out["forced_one_rank_dp"] = "TIM_AMD_BENCH_FORCE_DP=1: the data-parallel path on a one-rank RCCL group (every collective a copy) - a test mode, not a measurement"
    if gstep is not None and getattr(gstep, "kernel_nodes", None) is not None:
        # counted, not assumed: the kernel nodes of the captured step (hipGraphGetNodes on the graph torch captured)
        out["launches_per_step"] = {"kernel_nodes": gstep.kernel_nodes, "all_nodes": gstep.nodes,
                                    "source": "hipGraphGetNodes / hipGraphNodeGetType on the captured step"}
        if eager is not None:
            eager["launches_per_step"] = gstep.kernel_nodes
    if eager is not None:
        out["eager"] = eager
    if fam is not None and (fam[1][2] or fam[2][2]):
        # attention and LayerNorm launches of the SAME bracketed eager step as the roofline's GEMM events (HIP events on the launch
        # stream; each bracket includes the ~5 us between an event and its kernel, like the GEMM ones)
        gem, att, lnf = fam
        tot_ms = dt / args.steps * 1e3
        out["non_gemm"] = {
            "attention": {"us_per_step": round(att[0] * 1e3, 1), "launches": att[2],
                          "algorithmic_TBps": round(att[1] / att[0] / 1e9, 2) if att[0] > 0 else None},
            "layernorm": {"us_per_step": round(lnf[0] * 1e3, 1), "launches": lnf[2],
                          "algorithmic_TBps": round(lnf[1] / lnf[0] / 1e9, 2) if lnf[0] > 0 else None},
            "gemm_us_per_step": round(gem[0] * 1e3, 1), "gemm_launches": gem[2],
            "non_gemm_us_per_step": round((tot_ms - gem[0]) * 1e3, 1),
            "note": "HIP-event brackets inside one eager step (attention: qkv / o / dO / dqkv bytes; LayerNorm: rows read + written, "
                    "all LayerNorm launches incl. embedders and time MLP); non_gemm_us_per_step = timed ms_per_step - the GEMM brackets"}
    if comm is not None:
        out["comm"] = comm
    if fwd_ms:
        out["forward_only"] = {"ms_per_step": round(fwd_ms, 3), "interval_queries_per_s": round(B * (nv + na) / fwd_ms * 1e3, 1)}
    if gq is not None:
        peak = PEAK_BF16_TFLOPS if args.precision in ("bf16", "fp16") else PEAK_FP32_TFLOPS
        out["whole_step"] = {"tflops_algorithmic": round(value / world * gq / 1e3, 1),
                             "frac_of_mfma_peak": round(value / world * gq / 1e3 / peak, 4)}
    if rank == 0 and not args.no_roofline:
        per, fl, ms = ([], 0.0, 1.0) if args.no_per_shape else gemm_roofline(model, cfg, B, nv, na, args.precision)
        peak = PEAK_BF16_TFLOPS if args.precision in ("bf16", "fp16") else PEAK_FP32_TFLOPS
        ach_iso = fl / ms / 1e9
        ach = live[1] / live[0] / 1e9 if live and live[0] > 0 else ach_iso
        # algorithmic bytes per launch (operands + results + what the fused epilogue reads), C2a B=64 16-bit modes:
        #   8 NT launches per layer: sum over (M K + N K) * 2 B + outputs / residual / aux as each epilogue moves them
        #   1 grouped TN launch: dY 142.2 MB + X 101.6 MB (16-bit) + dW 33.5 MB (fp32) = 277.3 MB
        S_ = cfg.F + cfg.num_queries(nv, na)
        M_, E_, FF_ = B * S_, cfg.E, cfg.FF
        nt = [(3 * E_, E_, 2 * 3 * E_), (E_, E_, 8 * E_), (FF_, E_, 2 * 2 * FF_ + FF_ // 8), (E_, FF_, 8 * E_),      # forward
              (FF_, E_, 4 * FF_), (E_, FF_, 2 * E_), (E_, E_, 2 * E_), (E_, 3 * E_, 2 * E_)]                       # input gradients
        alg_nt = [M_ * k * 2 + n * k * 2 + M_ * ob for n, k, ob in nt]
        alg_tn = M_ * (E_ + FF_ + E_ + 3 * E_) * 2 + M_ * (FF_ + E_ + E_ + E_) * 2 + (2 * E_ * FF_ + 4 * E_ * E_) * 4
        # round 6: where the weight gradients of TWO layers go out as one launch (wgrad_p8_kernel, timhip_layer_bwd_weights_pair) the
        # family is 16 NT + 1 TN launch per pair of layers: 51 launches of the 6-layer stack instead of 54
        paired_tn = False
        try:
            import ctypes as _C
            from tim_amd import _lib as L_
            _d = L_.TimDesc(B, S_, cfg.F, cfg.d_model, E_, cfg.nhead, FF_, model.rt.prec, 0.1, 0, 0, 0, None)
            paired_tn = (cfg.num_layers >= 2 and not model.rt.overlap_wgrad and os.environ.get("TIM_AMD_WGRAD_PAIR", "1") != "0"
                         and L_.load().timhip_layer_wgrad_pair_wins(_C.byref(_d)) == 1)
        except Exception:  # noqa: BLE001
            paired_tn = False
        alg_avg = (2 * sum(alg_nt) + 2 * alg_tn) / 17.0 if paired_tn else (sum(alg_nt) + alg_tn) / 9.0
        traffic = None  # HBM-side bytes per launch of the GEMM kernels, from the committed rocprofv3 PMC passes
        tnote = "no PMC summary committed for this configuration"
        try:
            tfile = [f for f in ("r06_am_pmc_traffic.json", "r06_ag_pmc_traffic.json", "r06_ac_pmc_traffic.json", "r06_w_pmc_traffic.json", "r06_q_pmc_traffic.json", "r06_m_pmc_traffic.json", "r06_g_pmc_traffic.json", "r05_k_pmc_traffic.json", "r05_j_pmc_traffic.json", "r04_pmc_traffic.json", "r03_g_pmc_traffic.json", "r03_f_pmc_traffic.json", "r03_e_pmc_traffic.json", "r03_d_pmc_traffic.json", "r03_c_pmc_traffic.json", "r03_a_pmc_traffic.json", "r02_pmc_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", f))][0]
            tj = json.load(open(os.path.join(ROOT, "profiles", tfile)))["kernels"]
            if args.precision in ("bf16", "fp16") and args.workload == "C2a" and B == 64:
                ntk, tnk = tj.get("gemm_nt_ld_kernel", tj.get("gemm_nt_pp_kernel")), tj.get("wgrad_ld_kernel", tj.get("wgrad_pp_kernel"))   # 8 NT + 1 grouped TN launch per layer
                # (of a layer's eight NT launches five run one round of tiles - gemm_nt_ld - and three are multi-round: the tile walk
                #  gemm_nt_ldp and, from round 6, the eight-phase gemm_nt_p8 for two of them)
                multi = [tj[k] for k in ("gemm_nt_ldp_kernel", "gemm_nt_p8_kernel") if k in tj]
                if multi:
                    w = [m.get("launches", 1) for m in multi]
                    mb = sum(m["bytes_per_launch"] * wi for m, wi in zip(multi, w)) / max(sum(w), 1)
                    ntk = dict(ntk, bytes_per_launch=(5 * ntk["bytes_per_launch"] + 3 * mb) / 8.0)
                if paired_tn:
                    tnk = tj["wgrad_p8_kernel"]   # (KeyError -> no citation: a PMC set from before the paired launch)
                    traffic = round((16 * ntk["bytes_per_launch"] + tnk["bytes_per_launch"]) / 17.0)
                else:
                    traffic = round((8 * ntk["bytes_per_launch"] + tnk["bytes_per_launch"]) / 9.0)
                tnote = ("CITED, not measured in this run: average fabric-side bytes per GEMM launch from the committed rocprofv3 PMC "
                         "passes of this command on the builder's box (FETCH_SIZE x2 + WRITE_SIZE in separate passes, "
                         "profiles/%s, tools/pmc_traffic.py; the x2 and the write counter calibrated on known byte counts, "
                         "profiles/r04_pmc_calibration_notes.txt: both count Infinity-Cache hits, i.e. L2 <-> fabric traffic, an upper "
                         "bound of the HBM bytes): NT %.0f MB, grouped TN %.0f MB per launch"
                         % (tfile, ntk["bytes_per_launch"] / 1e6, tnk["bytes_per_launch"] / 1e6))
        except Exception:
            traffic = None
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": "cited" if traffic else None,
                           "algorithmic_bytes_per_launch": round(alg_avg),
                           "traffic_note": tnote + "; algorithmic bytes average %.0f MB per launch (NT %.0f MB on average, grouped "
                                           "TN %.1f MB%s)" % (alg_avg / 1e6, sum(alg_nt) / 8e6, alg_tn * (2 if paired_tn else 1) / 1e6,
                                                              " = two layers per launch" if paired_tn else ""),
                           "kernel": "MFMA GEMM family: gemm_nt_ld_kernel (160 x 256 tiles, loader waves + L2 prefetch: the one-round shapes) / gemm_nt_p8_kernel (round 6: 256 / 320 x 256 tiles on the eight-phase schedule for the in-projection and linear1 forward; TIMHIP_GEMM_P8=0: the tile walk gemm_nt_ldp_kernel) / gemm_nt_%s_kernel (8 launches per layer) + the grouped TN weight-"
                                     "gradient launch (round 6: wgrad_p8_kernel, the 8 weight gradients of TWO layers as one round of 256 x 256 eight-phase tiles; "
                                     "TIM_AMD_WGRAD_PAIR=0: wgrad_ld_kernel, a layer's 4 as one grid) = the "
                                     "72 GEMMs of the 6 encoder layers fwd+bwd in 51 (54) launches, 2*M*N*K algorithmic FLOPs each; `achieved` = sum FLOPs / sum of their HIP-event durations inside "
                                     "the last timed step, events recorded on the stream each kernel is launched on (default: the "
                                     "whole backward on one stream; TIM_AMD_OVERLAP_WGRAD=1 moves the weight gradients to a second "
                                     "stream, which lengthens every launch); `achieved_other_streams` = the same events on one "
                                     "extra step after the timed region in the other of the two stream configurations; "
                                     "`achieved_isolated` / `per_shape_isolated` = the same shapes timed back to back on an "
                                     "otherwise idle GPU" % ("h16" if args.precision in ("bf16", "fp16") else "f32"),
                           "events_from": ("the last of %d eager steps issued right before the capture and the timed region (a replayed graph cannot carry "
                                           "per-launch events; same kernels, launch parameters and stream)" % eager["steps"]) if eager
                                          else "the last timed step",
                           "backward_streams": 2 if model.rt.overlap_wgrad else 1,
                           "launches_timed": live[2] if live else 0,
                           "achieved_other_streams": round(live_serial, 1) if live and live_serial else None,
                           "achieved_isolated": round(ach_iso, 1), "per_shape_isolated": per}
    # the same step captured once in a HIP graph and replayed (tim_amd/graph.py): one host call per step.  Reported beside
    # `value`, which stays the eager number so that the roofline events above sit inside the timed region.  Measured in a CHILD
    # process (`--graph-child`): whatever a runtime does with a capture it dislikes - an exception, or a crash - the bench
    # line of this process is not at stake.
    if rank == 0 and not dp_on and not args.no_graph and gstep is None:
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--graph-child", "--workload", args.workload, "--batch", str(B),
               "--precision", args.precision, "--steps", str(args.steps), "--warmup", str(args.warmup)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out["graph_replay"] = json.loads(line[-1]) if line else {"error": "child exited with %d" % r.returncode}
        except Exception as e:  # noqa: BLE001
            out["graph_replay"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if rank == 0 and not dp_on and not detection and not args.no_secondary:
        # (1) the accuracy the headline is bought with, measured here: the timed model against the fp32 CPU oracle
        try:
            err, nlog = logit_parity(model, cfg, sd_np, nv, na, dev)
            out["parity"] = {"max_abs_logit_err": float("%.3g" % err), "vs": "fp32 oracle (oracle/tim_oracle.py on the host)",
                             "mode": args.precision, "sample": "%d logits of 4 synthetic windows, eval mode" % nlog,
                             "bound": 1e-3 if args.precision in ("fp16", "bf16", "bf16x3") else 1e-5}
        except Exception as e:  # noqa: BLE001
            out["parity"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        # (2) the plain-bf16 mode beside the fp16 headline (same build, same batch)
        if args.precision == "fp16":
            try:
                m2, _ = build_model(cfg, "bf16", dev, seed=0)
                m2.train()
                R2 = [None]   # (median of per-step event intervals after 30 warm-ups: a fresh model's first eager steps are host-bound)
                ms2 = robust_step_ms(lambda: step_fn(m2, batch, nv, na, R2), max(5, args.steps // 2), 30)[0]
                err2, _ = logit_parity(m2, cfg, sd_np, nv, na, dev)
                out["bf16_mode"] = {"ms_per_step": round(ms2, 3), "interval_queries_per_s": round(B * (nv + na) / ms2 * 1e3, 1),
                                    "max_abs_logit_err": float("%.3g" % err2),
                                    "note": "bf16 operands miss north_star's 1e-3 logit bound by ~10x; not the headline"}
                del m2
            except Exception as e:  # noqa: BLE001
                out["bf16_mode"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        # (3) one block per further BASELINE.json configuration (c2b: the window BASELINE.json words, 75 + 75 feature tokens;
        #     c1: the small visual-only model, eager and as a HIP-graph replay - its eager step is host-bound; c3: Perception
        #     Test; c4_train: detection as TRUE training - .train(), 399 drawn queries, on-device IoU labelling, focal + DIoU)
        if args.workload == "C2a":
            sec_steps = max(5, args.steps // 2)
            for key, wl, b_, kw, desc in (
                    ("c2a_b8", "C2a", 8, {"graph": True},
                     "C2a at 8 windows per GPU: the reference recipe (global batch 64, utils/parser.py:87) on 8 GPUs - datasets/loader.py:48 "
                     "divides the batch by the GPU count (M = 1240 rows); fixed-cotangent forward + backward like the headline"),
                    ("c2a_train", "C2a", B, {"graph": True, "rec_train": True},
                     "C2a, the reference's TRAINING ITERATION (recognition/scripts/train.py:184-366): time MLP -> mixup -> encoder -> "
                     "mixup cross entropy x 4 + cross-modal DRLoc -> backward -> clip_grad_norm_(1.0) -> AdamW(fused) -> weight refresh"),
                    ("c2b", "C2b", B, {"graph": True}, "C2b: 75+75 feature tokens, 15+10 queries (S = 205), train-mode dropout"),
                    ("c1", "C1", B, {"graph": True}, "C1: visual-only, d_model 256, 2 layers, 4 heads, 50 tokens + 3 x 10 queries (S = 80), train-mode dropout"),
                    ("c3", "C3", B, {"graph": True}, "C3: Perception Test A+V recognition, 50+50 tokens, 15+10 queries, dropouts 0.1"),
                    ("c4_train", "C4", 16, {"det_train": True, "graph": True},
                     "C4: EPIC-100 detection TRAINING step (det scripts/train.py:212-349): model.train(), 399 queries drawn from the "
                     "training pyramid, IoU labelling on the device, encoder forward, sigmoid focal loss with IoU row weights + 1-D "
                     "DIoU over the positives, full backward (S = 499)")):
                try:
                    blk = secondary_block(wl, b_, args.precision, dev, sec_steps, 3, **kw)
                    blk["workload"] = desc
                    out[key] = blk
                except Exception as e:  # noqa: BLE001
                    out[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if rank == 0 and not dp_on and not args.no_cpu_baseline and not detection:
        out["cpu_baseline"] = cpu_baseline(cfg, sd_np, nv, na, args.workload)
    if rank == 0:
        print(json.dumps(out))
    if dp_on:
        import torch.distributed as dist
        dist.barrier()       # rank 0 is the last to arrive (roofline loop): leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
