"""Detection query labelling on the device (csrc/labels.hip through tim_amd.detection.TIM.label_queries) against the vectors
the reference's own TIM.label_queries produced (tests/golden/labels_*.npz) and against the oracle on a full-size batch:
bit-exact fp32 IoUs / targets / smoothed label matrices."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tim_oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.test_labels_oracle import CASES, expected  # noqa: E402

DEV = "cuda:0"


def det_model(num_class, vn, thr, ls):
    from tim_amd.detection import TIM
    cfg = H.tiny_cfg("detection", "audio_visual", "audio_visual", vn, num_class=num_class)
    return TIM(cfg.num_class, visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim, d_model=cfg.d_model,
               nhead=cfg.nhead, num_layers=cfg.num_layers, num_feats=cfg.num_feats, include_verb_noun=vn, iou_threshold=thr,
               label_smoothing=ls, precision="fp32").to(DEV)


def target_of(g):
    return {k[7:]: torch.from_numpy(g[k]).to(DEV) for k in g.files if k.startswith("target/")}


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[7:-4] for p in CASES])
def test_label_queries_matches_reference_vectors(path):
    g = np.load(path)
    nc = g["num_class"]
    vn = bool(int(g["vn"]))
    num_class = [[int(nc[0]), int(nc[1]), int(nc[2])], int(nc[3])] if vn else (int(nc[2]), int(nc[3]))
    m = det_model(num_class, vn, float(g["iou_threshold"]), float(g["label_smoothing"]))
    q = torch.from_numpy(g["queries"]).to(DEV)
    for mod in ("visual", "audio"):
        tg, labels, ious = m.label_queries(q.clone(), target_of(g), mod, m.iou_threshold)
        torch.cuda.synchronize()
        etg, emats, eious = expected(g, mod)
        assert np.array_equal(ious.cpu().numpy(), eious), mod
        assert np.array_equal(tg.cpu().numpy(), etg), mod
        mats = [t for t in labels if t.numel()] if mod == "visual" else [labels]
        assert len(mats) == len(emats)
        for a, b in zip(mats, emats):
            assert tuple(a.shape) == tuple(b.shape) and np.array_equal(a.cpu().numpy(), b), mod
        if mod == "visual" and not vn:   # the reference returns empty verb / noun placeholders (tim.py:159-160)
            assert labels[0].numel() == 0 and labels[1].numel() == 0


def test_label_queries_full_size_vs_oracle():
    """EPIC-100 detection sizes: 16 windows x 399 queries, 97 verb / 300 noun / 3806 action / 44 audio classes"""
    rng = np.random.default_rng(3)
    B, Ng = 16, 12
    m = det_model([[97, 300, 3806], 44], True, 0.6, 0.9)
    q = m.inference_queries.repeat(B, 1, 1).to(DEV)
    st = rng.uniform(0, 0.9, size=(B, Ng)).astype(np.float32)
    segs = np.round(np.stack([st, np.minimum(st + rng.choice([0.01, 0.02, 0.04, 0.08, 0.16], size=(B, Ng)), 1.0)], -1), 3).astype(np.float32)
    segs[:, Ng // 2:] = 0.0   # padding slots
    lab = lambda hi: torch.from_numpy(np.where(np.arange(Ng)[None] < Ng // 2, rng.integers(0, hi, size=(B, Ng)), -1).astype(np.int64))
    target = {"v_gt_segments": torch.from_numpy(segs), "a_gt_segments": torch.from_numpy(segs), "verb": lab(97), "noun": lab(300),
              "action": lab(3806), "class_id": lab(44)}
    tg, labels, ious = m.label_queries(q, {k: v.to(DEV) for k, v in target.items()}, "visual", 0.6)
    torch.cuda.synchronize()
    gl = torch.stack([target["verb"], target["noun"], target["action"]], -1)
    etg, emats, eious = O.label_queries(q.cpu(), target["v_gt_segments"], gl, 0.6, 0.9, [97, 300, 3806])
    assert torch.equal(ious.cpu(), eious) and torch.equal(tg.cpu(), etg)
    assert 0 < int(torch.isfinite(etg[:, 0]).sum()) < etg.shape[0]
    for a, b in zip(labels, emats):
        assert torch.equal(a.cpu(), b)
