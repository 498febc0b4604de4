"""The oracle's detection query labelling against vectors produced by the reference's own TIM.label_queries
(tests/golden/make_golden_r2.py; det tim.py:157-270): bit-exact (fp32 interval arithmetic, int64 index work)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import tim_oracle as O
from tests import helpers as H

CASES = sorted(glob.glob(os.path.join(H.GOLDEN, "labels_*.npz")))


def oracle_labels(g, mod):
    q = torch.from_numpy(g["queries"])
    nc = g["num_class"]
    if mod == "visual":
        segs = torch.from_numpy(g["target/v_gt_segments"])
        lab = torch.stack([torch.from_numpy(g["target/" + k]) for k in ("verb", "noun", "action")], -1)
        counts = [int(nc[0]), int(nc[1]), int(nc[2])] if int(g["vn"]) else [int(nc[2])]
        if not int(g["vn"]):
            lab = lab[..., 2:]
    else:
        segs = torch.from_numpy(g["target/a_gt_segments"])
        lab = torch.from_numpy(g["target/class_id"])[..., None]
        counts = [int(nc[3])]
    return O.label_queries(q, segs, lab, float(g["iou_threshold"]), float(g["label_smoothing"]), counts)


def expected(g, mod):
    if mod == "visual":
        names = ("verb", "noun", "action") if int(g["vn"]) else ("action",)
        return g["visual/targets"], [g["visual/labels_" + n] for n in names], g["visual/ious"]
    return g["audio/targets"], [g["audio/labels"]], g["audio/ious"]


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[7:-4] for p in CASES])
@pytest.mark.parametrize("mod", ["visual", "audio"])
def test_label_queries_matches_reference(path, mod):
    g = np.load(path)
    tg, mats, ious = oracle_labels(g, mod)
    etg, emats, eious = expected(g, mod)
    assert np.array_equal(ious.numpy(), eious)
    assert np.array_equal(tg.numpy(), etg)                      # inf == inf
    assert len(mats) == len(emats)
    for a, b in zip(mats, emats):
        assert a.shape == tuple(b.shape) and np.array_equal(a.numpy(), b)


def test_fixtures_cover_the_edge_cases():
    neg = np.load(os.path.join(H.GOLDEN, "labels_negstart_vn.npz"))
    assert (neg["target/v_gt_segments"][..., 0] < 0).any()      # the offset path of get_query_ious
    assert np.isfinite(neg["visual/targets"]).any()
    inf = np.load(os.path.join(H.GOLDEN, "labels_inference_vn.npz"))
    assert (inf["visual/ious"] == 1.0).any()                    # a segment that is itself a pyramid query
