"""Golden vectors for the loss tail from the reference's own functions (build container only).

    python tests/golden/make_golden_loss.py              (recognition: mixup CE, DRLoc)
    python tests/golden/make_golden_loss.py detection    (detection: focal, DIoU)

Imports utils/mixup.py and models/helpers/losses/drloc.py of /root/reference/recognition (torch + numpy only) and the
reference TIM module (for its drloc_mlp), feeds them inputs that `tim_amd.synth` regenerates from a seed, and stores
inputs' seeds, the sampled positions, losses and gradients as plain numbers in tests/golden/loss_*.npz.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sj = types.ModuleType("simplejson")
sj.dumps = lambda *a, **k: ""
sys.modules["simplejson"] = sj
for name in ("fvcore", "fvcore.common", "fvcore.common.file_io"):
    sys.modules[name] = types.ModuleType(name)


class _PM:
    open = staticmethod(open)


sys.modules["fvcore.common.file_io"].PathManager = _PM
VARIANT = "detection" if len(sys.argv) > 1 and sys.argv[1] == "detection" else "recognition"
sys.path.insert(0, "/root/reference/" + VARIANT)
from time_interval_machine.models.tim import TIM  # noqa: E402
if VARIANT == "recognition":
    from time_interval_machine.utils.mixup import mixup_criterion  # noqa: E402
import time_interval_machine.models.helpers.losses.drloc as ref_drloc  # noqa: E402

from tim_amd import synth  # noqa: E402
from tim_amd.config import named_config  # noqa: E402

torch.set_num_threads(8)


def ce_case(name, rows, C, lam, seed, frac_invalid):
    """train.py:218-258 for one head: filter the valid rows of both target sets, criterion twice, blend."""
    logits = torch.from_numpy(synth.normal(seed, "ce_logits", (rows, C), std=2.0)).double().requires_grad_(True)
    u = synth.uniform01(seed, "ce_targets", (rows, 4))
    ya = torch.from_numpy(np.floor(u[:, 0] * C).astype(np.int64))
    yb = torch.from_numpy(np.floor(u[:, 1] * C).astype(np.int64))
    ya[torch.from_numpy(u[:, 2] < frac_invalid)] = -1
    yb[torch.from_numpy(u[:, 3] < frac_invalid)] = -1
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.2, ignore_index=-1)
    va, vb = ya != -1, yb != -1
    loss = mixup_criterion(crit, logits[va], logits[vb], ya[va], yb[vb], lam)
    loss.backward()
    g = logits.grad.numpy()
    # big heads: keep the first 128 columns plus per-row sums (the full gradient is checked against the oracle)
    np.savez(os.path.join(HERE, "loss_ce_%s.npz" % name), rows=rows, C=C, lam=lam, seed=seed, ya=ya.numpy(),
             yb=yb.numpy(), loss=loss.item(), dlogits=g[:, :128].astype(np.float32), row_abs=np.abs(g).sum(1),
             row_sum=g.sum(1))
    print(name, "loss", loss.item())


def drloc_case(name, cfg_name, n, m, seed, crossmodal):
    cfg = named_config(cfg_name)
    ref = TIM(cfg.num_class, visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim,
              feat_drop=cfg.feat_drop, seq_drop=cfg.seq_drop, d_model=cfg.d_model, feedforward_scale=cfg.feedforward_scale,
              nhead=cfg.nhead, num_layers=cfg.num_layers, enc_dropout=cfg.enc_dropout, input_modality=cfg.input_modality,
              data_modality=cfg.data_modality, num_feats=cfg.num_feats, include_verb_noun=cfg.include_verb_noun)
    sd = synth.make_state_dict(cfg, seed=seed, dtype=np.float64)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ref = ref.double().eval()
    E, F = cfg.E, cfg.F
    feats = torch.from_numpy(synth.normal(seed, "drloc_feats", (n, F, E))).double().requires_grad_(True)
    l = cfg.num_feats if crossmodal else F
    g = torch.Generator().manual_seed(seed)
    pos_1 = torch.randint(l, size=(n, m), generator=g)
    pos_2 = torch.randint(l, size=(n, m), generator=g)
    ref_drloc.position_sampling = lambda k, mm, nn: (pos_1, pos_2)   # pin the drawn pairs
    if crossmodal:
        loss = ref_drloc.dense_relative_localization_loss_crossmodal(feats[:, :l], feats[:, l:], ref, m)
    else:
        loss = ref_drloc.dense_relative_localization_loss(feats, ref, m)
    loss.backward()
    grads = {"g_" + k: p.grad.numpy().astype(np.float32) for k, p in ref.named_parameters() if k.startswith("drloc_mlp.")}
    np.savez(os.path.join(HERE, "loss_drloc_%s.npz" % name), cfg=cfg_name, n=n, m=m, seed=seed, crossmodal=crossmodal,
             pos_1=pos_1.numpy(), pos_2=pos_2.numpy(), loss=loss.item(), dfeats=feats.grad.numpy().astype(np.float32),
             **grads)
    print(name, "loss", loss.item())


def det_case(name, rows, C, n_reg, seed):
    """detection: get_loss(sigmoid_focal_loss, preds[valid], targets[valid], weights=ious, reduction="sum") and
    get_loss(ctr_diou_loss_1d, reg[pos], offsets[pos], reduction="sum") as det train.py:222-285 calls them.
    Run in a separate process: python tests/golden/make_golden_loss.py detection"""
    from time_interval_machine.models.helpers.losses.sigmoid import sigmoid_focal_loss
    from time_interval_machine.models.helpers.losses.iou import ctr_diou_loss_1d
    from time_interval_machine.models.helpers.losses.loss import get_loss
    logits = torch.from_numpy(synth.normal(seed, "focal_logits", (rows, C), std=3.0)).float().requires_grad_(True)
    u = synth.uniform01(seed, "focal_aux", (rows, 3))
    cls = np.floor(u[:, 0] * C).astype(np.int64)
    tgt = np.full((rows, C), 0.1 / C, dtype=np.float32)                 # smoothed one-hot (det tim.py:157-184)
    tgt[np.arange(rows), cls] += 0.9
    tgt[u[:, 1] < 0.3] = 0.0                                              # background rows
    targets = torch.from_numpy(tgt)
    ious = torch.from_numpy((u[:, 2] * 1.2 - 0.2).astype(np.float32))    # < 0: row not used for classification
    valid = ious >= 0.0
    w = ious[valid].clone()
    w.masked_fill_(w < 0.6, 1.0)
    loss = get_loss(sigmoid_focal_loss, logits[valid], targets[valid], weights=w, reduction="sum")
    loss.backward()
    elem = get_loss(sigmoid_focal_loss, logits[valid].detach(), targets[valid], weights=w, reduction="none")
    v = synth.uniform(seed, "diou", (n_reg, 4), 0.0, 1.0).astype(np.float32)
    # (no exact ties / all-zero rows: there the reference's own gradient depends on whether TorchScript is still in its
    #  profiling run - eager tie-splitting - or already runs the differentiated graph - strict comparisons)
    pred = torch.from_numpy(v[:, :2].copy()).requires_grad_(True)
    off = torch.from_numpy(v[:, 2:].copy())
    reg = get_loss(ctr_diou_loss_1d, pred, off, reduction="sum")
    reg.backward()
    np.savez(os.path.join(HERE, "loss_det_%s.npz" % name), rows=rows, C=C, seed=seed, targets=tgt, ious=ious.numpy(),
             focal=loss.item(), dlogits=logits.grad.numpy(), elem_rowsum=elem.sum(1).numpy(), pred=v[:, :2], off=v[:, 2:],
             diou=reg.item(), dpred=pred.grad.numpy())
    print(name, "focal", loss.item(), "diou", reg.item())


if __name__ == "__main__" and VARIANT == "detection":
    det_case("small", 29, 13, 9, 31)
    det_case("verb", 399 * 2, 97, 40, 32)

if __name__ == "__main__" and VARIANT == "recognition":
    ce_case("small", 37, 13, 0.3, 11, 0.25)
    ce_case("action", 96, 3806, 0.71, 12, 0.3)
    ce_case("nomix", 50, 97, 1.0, 13, 0.2)
    drloc_case("tiny_cross", "tiny", 3, 5, 21, True)
    drloc_case("tiny_single", "tiny", 3, 7, 22, False)
