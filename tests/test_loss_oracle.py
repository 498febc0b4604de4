"""Loss tail: the oracle's restatement of the mixup criterion and of DRLoc against golden vectors produced
by the reference's own functions (tests/golden/make_golden_loss.py).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import tim_oracle as O
from tim_amd import synth
from tim_amd.config import named_config
from tests.helpers import GOLDEN

CE_CASES = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "loss_ce_*.npz")))
DET_CASES = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "loss_det_*.npz")))
DR_CASES = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "loss_drloc_*.npz")))


def ce_inputs(g, dtype=torch.float64):
    rows, C, seed = int(g["rows"]), int(g["C"]), int(g["seed"])
    logits = torch.from_numpy(synth.normal(seed, "ce_logits", (rows, C), std=2.0)).to(dtype)
    return logits, torch.from_numpy(g["ya"]), torch.from_numpy(g["yb"]), float(g["lam"])


def drloc_inputs(g, dtype=torch.float64):
    cfg = named_config(str(g["cfg"]))
    n, seed = int(g["n"]), int(g["seed"])
    sd = {k: torch.from_numpy(v).to(dtype) for k, v in synth.make_state_dict(cfg, seed=seed, dtype=np.float64).items()}
    feats = torch.from_numpy(synth.normal(seed, "drloc_feats", (n, cfg.F, cfg.E))).to(dtype)
    return cfg, sd, feats, torch.from_numpy(g["pos_1"]), torch.from_numpy(g["pos_2"])


@pytest.mark.parametrize("case", CE_CASES)
def test_mixup_ce_oracle_matches_reference(case):
    g = np.load(os.path.join(GOLDEN, case))
    logits, ya, yb, lam = ce_inputs(g)
    logits.requires_grad_(True)
    loss = O.mixup_ce(logits, ya, yb, lam, 0.2)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-12
    d = logits.grad.numpy()
    assert np.abs(d[:, :128] - g["dlogits"]).max() < 1e-7
    assert np.abs(np.abs(d).sum(1) - g["row_abs"]).max() < 1e-10
    assert np.abs(d.sum(1) - g["row_sum"]).max() < 1e-10


@pytest.mark.parametrize("case", DR_CASES)
def test_drloc_oracle_matches_reference(case):
    g = np.load(os.path.join(GOLDEN, case))
    cfg, sd, feats, p1, p2 = drloc_inputs(g)
    feats.requires_grad_(True)
    for k in list(sd):
        if k.startswith("drloc_mlp."):
            sd[k].requires_grad_(True)
    l = cfg.num_feats if bool(g["crossmodal"]) else cfg.F
    x1, x2 = (feats[:, :l], feats[:, l:]) if bool(g["crossmodal"]) else (feats, feats)
    loss = O.drloc_loss(sd, x1, x2, p1, p2)
    loss.backward()
    # the reference's F.l1_loss(fp32 deltax, fp64 pred) returns an fp32 scalar: compare at fp32 resolution
    assert abs(loss.item() - float(g["loss"])) < 6e-8
    assert np.abs(feats.grad.numpy() - g["dfeats"]).max() < 1e-7
    for k in sd:
        if k.startswith("drloc_mlp."):
            assert np.abs(sd[k].grad.numpy() - g["g_" + k]).max() < 1e-7, k


def det_inputs(g):
    rows, C, seed = int(g["rows"]), int(g["C"]), int(g["seed"])
    logits = torch.from_numpy(synth.normal(seed, "focal_logits", (rows, C), std=3.0)).float()
    ious = torch.from_numpy(g["ious"])
    valid = ious >= 0.0
    w = ious.clone()
    w[(w < 0.6) & valid] = 1.0          # det train.py:228: v_ious.masked_fill_(v_ious < iou_threshold, 1.0) on the valid rows
    return logits, torch.from_numpy(g["targets"]), w, valid


@pytest.mark.parametrize("case", DET_CASES)
def test_detection_losses_oracle_matches_reference(case):
    g = np.load(os.path.join(GOLDEN, case))
    logits, targets, w, valid = det_inputs(g)
    x = logits.clone().requires_grad_(True)
    loss = O.focal_loss(x[valid], targets[valid], w[valid])
    loss.backward()
    assert abs(loss.item() - float(g["focal"])) <= 1e-6 * abs(float(g["focal"]))
    assert np.abs(x.grad.numpy() - g["dlogits"]).max() <= 1e-6 * np.abs(g["dlogits"]).max()
    elem = O.focal_loss(logits[valid], targets[valid], w[valid], reduction="none")
    assert np.abs(elem.sum(1).numpy() - g["elem_rowsum"]).max() <= 1e-5 * np.abs(g["elem_rowsum"]).max()
    pred = torch.from_numpy(g["pred"]).requires_grad_(True)
    reg = O.diou_1d(pred, torch.from_numpy(g["off"]))
    reg.backward()
    assert abs(reg.item() - float(g["diou"])) <= 1e-6 * abs(float(g["diou"]))
    assert np.abs(pred.grad.numpy() - g["dpred"]).max() <= 1e-6
