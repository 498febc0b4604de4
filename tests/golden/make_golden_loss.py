"""Golden vectors for the loss tail from the reference's own functions (build container only).

    python tests/golden/make_golden_loss.py

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
sys.path.insert(0, "/root/reference/recognition")
from time_interval_machine.models.tim import TIM  # noqa: E402
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


if __name__ == "__main__":
    ce_case("small", 37, 13, 0.3, 11, 0.25)
    ce_case("action", 96, 3806, 0.71, 12, 0.3)
    ce_case("nomix", 50, 97, 1.0, 13, 0.2)
    drloc_case("tiny_cross", "tiny", 3, 5, 21, True)
    drloc_case("tiny_single", "tiny", 3, 7, 22, False)
