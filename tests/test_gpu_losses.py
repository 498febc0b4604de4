"""Loss tail on the HIP kernels (tim_amd/losses.py -> tim_amd/csrc/losses.hip + the GEMM kernels) against the oracle
and the reference-generated golden vectors.  fp32 precision: 1e-5 relative to the gradient scale; bf16: 1e-2 (three
bf16 GEMMs deep, compared with the fp32 oracle)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tim_oracle as O  # noqa: E402
from tim_amd import losses  # noqa: E402
from tim_amd.tim import TIM  # noqa: E402
from tests.helpers import GOLDEN  # noqa: E402
from tests.test_loss_oracle import CE_CASES, DET_CASES, DR_CASES, ce_inputs, det_inputs, drloc_inputs  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("case", CE_CASES)
def test_mixup_cross_entropy(case):
    g = np.load(os.path.join(GOLDEN, case))
    logits, ya, yb, lam = ce_inputs(g, torch.float32)
    x = logits.to(DEV).requires_grad_(True)
    loss = losses.mixup_cross_entropy(x, ya.to(DEV), yb.to(DEV), lam, label_smoothing=0.2)
    (loss * 3.0).backward()          # a non-trivial upstream gradient
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) < 2e-6 * max(1.0, abs(float(g["loss"])))
    d = x.grad.cpu().numpy() / 3.0
    scale = float(np.abs(g["dlogits"]).max())
    assert np.abs(d[:, :128] - g["dlogits"]).max() < 1e-5 * scale
    assert np.abs(np.abs(d).sum(1) - g["row_abs"]).max() < 1e-4 * float(g["row_abs"].max())
    # the two-call form the reference spells out (criterion on the filtered rows, twice) gives the same value
    crit = losses.CrossEntropyLoss(label_smoothing=0.2, ignore_index=-1)
    va, vb = (ya != -1).to(DEV), (yb != -1).to(DEV)
    x2 = logits.to(DEV)
    two = losses.mixup_criterion(crit, x2[va], x2[vb], ya.to(DEV)[va], yb.to(DEV)[vb], lam)
    assert abs(two.item() - loss.item()) < 2e-6 * max(1.0, abs(loss.item()))


def test_cross_entropy_edge_cases():
    x = torch.randn(5, 7, device=DEV, requires_grad=True)
    y = torch.tensor([-1, -1, -1, -1, -1], device=DEV)
    loss = losses.CrossEntropyLoss(0.2)(x, y)          # nothing valid: 0 loss, 0 gradient
    loss.backward()
    assert loss.item() == 0.0 and float(x.grad.abs().max()) == 0.0
    ref = O.mixup_ce(x.detach().cpu().double(), torch.tensor([3, -1, 0, 6, -1]), None, 1.0, 0.0)
    got = losses.CrossEntropyLoss(0.0)(x.detach(), torch.tensor([3, -1, 0, 6, -1], device=DEV))
    assert abs(got.item() - ref.item()) < 1e-5


def _model(cfg, sd, prec):
    m = TIM(cfg.num_class, visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim,
            feat_drop=cfg.feat_drop, seq_drop=cfg.seq_drop, d_model=cfg.d_model, feedforward_scale=cfg.feedforward_scale,
            nhead=cfg.nhead, num_layers=cfg.num_layers, enc_dropout=cfg.enc_dropout, input_modality=cfg.input_modality,
            data_modality=cfg.data_modality, num_feats=cfg.num_feats, include_verb_noun=cfg.include_verb_noun,
            precision=prec)
    m.load_state_dict({k: v.float() for k, v in sd.items()})
    return m.to(DEV)


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", DR_CASES)
def test_drloc_loss(case, prec):
    g = np.load(os.path.join(GOLDEN, case))
    cfg, sd, feats, p1, p2 = drloc_inputs(g, torch.float64)
    model = _model(cfg, sd, prec)
    x = feats.float().to(DEV).requires_grad_(True)
    cross = bool(g["crossmodal"])
    m = int(g["m"])
    if cross:
        l = cfg.num_feats
        loss = losses.dense_relative_localization_loss_crossmodal(x[:, :l], x[:, l:], model, m, positions=(p1, p2))
    else:
        loss = losses.dense_relative_localization_loss(x, model, m, positions=(p1, p2))
    loss.backward()
    torch.cuda.synchronize()
    tol = {"fp32": 1e-5, "bf16": 1e-2, "fp16": 2e-3}[prec]   # fp16: the default mode; its gradient operands are device-scaled
    assert abs(loss.item() - float(g["loss"])) < tol * max(1.0, abs(float(g["loss"])))
    sc = float(np.abs(g["dfeats"]).max())
    assert np.abs(x.grad.cpu().numpy() - g["dfeats"]).max() < tol * sc * (1 if prec == "fp32" else 3)
    for k, p in model.named_parameters():
        if k.startswith("drloc_mlp."):
            ref = g["g_" + k]
            assert np.abs(p.grad.cpu().numpy() - ref).max() < tol * max(float(np.abs(ref).max()), 1e-3) * (1 if prec == "fp32" else 3), k
    # the module entry point the reference's drloc.py calls: model(cat(pts_1, pts_2), "drloc_mlp")
    pts = torch.cat([losses.collect_samples(x.detach()[:, :l] if cross else x.detach(), p1.to(DEV), x.shape[0]).transpose(1, 2),
                     losses.collect_samples(x.detach()[:, l:] if cross else x.detach(), p2.to(DEV), x.shape[0]).transpose(1, 2)],
                    dim=2)
    pred = model(pts, "drloc_mlp")
    ref = O.drloc_mlp({k: v.double() for k, v in sd.items()}, pts.cpu().double())
    assert pred.shape == ref.shape
    assert (pred.cpu().double() - ref).abs().max().item() < tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("case", DET_CASES)
def test_detection_losses(case):
    """focal (row weights, row filter, sum / none) and 1-D DIoU on the HIP kernels vs the reference-generated vectors"""
    g = np.load(os.path.join(GOLDEN, case))
    logits, targets, w, valid = det_inputs(g)
    x = logits.to(DEV).requires_grad_(True)
    # masked form: no row filtering on the host
    loss = losses.focal_loss_sum(x, targets.to(DEV), row_weights=w.to(DEV), row_valid=valid.to(DEV))
    (loss * 0.5).backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["focal"])) <= 2e-5 * abs(float(g["focal"]))
    sc = float(np.abs(g["dlogits"]).max())
    assert np.abs(x.grad.cpu().numpy() * 2.0 - g["dlogits"]).max() <= 1e-5 * sc
    # the reference's call: get_loss(criterion, preds[valid], targets[valid], weights=ious, reduction=...)
    xv, tv, wv = logits[valid].to(DEV), targets[valid].to(DEV), w[valid].to(DEV)
    same = losses.get_loss(losses.sigmoid_focal_loss, xv, tv, weights=wv, reduction="sum")
    assert abs(same.item() - float(g["focal"])) <= 2e-5 * abs(float(g["focal"]))
    elem = losses.get_loss(losses.sigmoid_focal_loss, xv, tv, weights=wv, reduction="none")
    assert elem.shape == xv.shape
    assert np.abs(elem.sum(1).cpu().numpy() - g["elem_rowsum"]).max() <= 2e-5 * np.abs(g["elem_rowsum"]).max()
    plain = losses.sigmoid_focal_loss(xv, tv, reduction="mean")
    ref = O.focal_loss(logits[valid].double(), targets[valid].double(), None, reduction="mean")
    assert abs(plain.item() - ref.item()) <= 2e-5 * abs(ref.item())
    # DIoU
    pred = torch.from_numpy(g["pred"]).to(DEV).requires_grad_(True)
    reg = losses.get_loss(losses.ctr_diou_loss_1d, pred, torch.from_numpy(g["off"]).to(DEV), reduction="sum")
    reg.backward()
    torch.cuda.synchronize()
    assert abs(reg.item() - float(g["diou"])) <= 1e-5 * abs(float(g["diou"]))
    assert np.abs(pred.grad.cpu().numpy() - g["dpred"]).max() <= 1e-4


@pytest.mark.parametrize("heads,npos", [(1, 40), (3, 40), (2, 0)])
def test_detection_side_loss_equals_its_composition(heads, npos):
    """losses.detection_side_loss (timhip_det_side_loss_fwd / _bwd: one modality side of det scripts/train.py:222-349 in a handful
    of launches, flags and weights derived inside the kernels) against the composition it replaces - focal_loss_sum /
    diou_loss_sum (pinned to the reference-generated vectors above) and the loop's torch glue in float64: loss value, the EMA
    normaliser it leaves behind, gradients of every head's logits and of the regression outputs; also without a single positive
    row (no regression term, normaliser advanced with max(0, 1)), and twice in a row (the normaliser is a running value)."""
    rows, thr, lam, mom = 300, 0.6, 0.5, 0.9
    g = torch.Generator().manual_seed(7 + heads)
    Cs = [13, 29, 97][:heads]
    logits = [torch.randn(rows, c, generator=g) * 2.0 for c in Cs]
    targets = [torch.rand(rows, c, generator=g).pow(8.0) for c in Cs]          # smoothed-label-like: mostly near 0
    iou = torch.rand(rows, generator=g)
    iou[torch.randperm(rows, generator=g)[:30]] = -1.0                          # rows outside every ground-truth window
    pos = torch.zeros(rows, dtype=torch.bool)
    cand = torch.nonzero(iou >= thr).flatten()
    pos[cand[:npos]] = True
    off = torch.full((rows, 2), float("inf"))
    off[pos] = torch.rand(int(pos.sum()), 2, generator=g) * 3.0
    reg = torch.rand(rows, 2, generator=g) * 3.0
    norm0 = 250.0

    def composed(norm_in):
        xs = [x.double().requires_grad_(True) for x in logits]
        r = reg.double().requires_grad_(True)
        w = torch.where(iou < thr, torch.ones_like(iou), iou).double()
        valid = iou >= 0
        num_pos = int(pos.sum())
        nm = mom * norm_in + (1.0 - mom) * max(num_pos, 1)
        cls = sum(O.focal_loss(x[valid], t.double()[valid], w[valid], reduction="sum") for x, t in zip(xs, targets)) / (heads * nm)
        tot = cls
        if num_pos > 0:
            tot = tot + lam * O.diou_1d(r[pos], off.double()[pos]).sum() / nm
        (tot * 0.7).backward()
        return tot.item(), nm, [x.grad for x in xs], (r.grad if r.grad is not None else torch.zeros_like(r))

    want1 = composed(norm0)
    want2 = composed(want1[1])
    norm = torch.full((), norm0, dtype=torch.float32, device=DEV)
    for want in (want1, want2):
        xs = [x.to(DEV).requires_grad_(True) for x in logits]
        r = reg.to(DEV).requires_grad_(True)
        loss = losses.detection_side_loss(xs, [t.to(DEV) for t in targets], r, off.to(DEV), iou.to(DEV), norm, thr, lambda_reg=lam,
                                          momentum=mom)
        (loss * 0.7).backward()
        torch.cuda.synchronize()
        assert abs(loss.item() - want[0]) <= 2e-5 * abs(want[0]), (loss.item(), want[0])
        assert abs(norm.item() - want[1]) <= 1e-5 * want[1]
        for x, gw in zip(xs, want[2]):
            sc = gw.abs().max().item()
            assert (x.grad.cpu().double() - gw).abs().max().item() <= 2e-5 * sc
            assert bool((x.grad.cpu()[iou < 0] == 0).all())
        gr = r.grad.cpu().double() if r.grad is not None else torch.zeros(rows, 2, dtype=torch.float64)
        assert (gr - want[3]).abs().max().item() <= 1e-5 * max(1e-6, want[3].abs().max().item()) + 1e-9
        assert bool((gr[~pos] == 0).all())
