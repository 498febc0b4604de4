"""Device-side batch assembly (tim_amd/data.py -> timhip_window_gather / timhip_window_times) against the reference's
`__getitem__` outputs and the oracle's collate.  Bit-exact (pure gathers and one IEEE subtraction / division)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import data_oracle as D  # noqa: E402
from tim_amd.data import DeviceWindowDataset  # noqa: E402
from tests.helpers import GOLDEN  # noqa: E402
from tests.golden.batch_inputs import make_tables  # noqa: E402
from tests.test_batch_oracle import CASES  # noqa: E402


@pytest.mark.parametrize("case", CASES)
def test_device_batch_matches_reference(case):
    g = np.load(os.path.join(GOLDEN, case))
    tb = make_tables(int(g["seed"]), str(g["modality"]))
    ds = DeviceWindowDataset(tb["windows"], tb["num_feats"], tb["window_size"], tb["max_visual_actions"],
                             tb["max_audio_actions"], tb["model_modality"], tb["v_feats"], tb["v_feat_times"], tb["a_feats"],
                             tb["a_feat_times"])
    idx = [5, 0, 3, 3, 6, 1, 2, 4]                      # repeated and out of order
    has_v, has_a = "visual" in tb["model_modality"], "audio" in tb["model_modality"]
    va = np.stack([g["va%d" % i] for i in idx]) if has_v else None
    aa = np.stack([g["aa%d" % i] for i in idx]) if has_a else None
    v, a, t, label, meta = ds.batch(idx, va, aa)
    torch.cuda.synchronize()
    # against the reference's own per-sample outputs
    for b, i in enumerate(idx):
        if has_v:
            assert np.array_equal(v[b].cpu().numpy(), g["v%d" % i])
        if has_a:
            assert np.array_equal(a[b].cpu().numpy(), g["a%d" % i])
        assert np.array_equal(t[b].cpu().numpy(), g["t%d" % i])
        for k in ("verb", "noun", "action", "class_id"):
            assert np.array_equal(label[k][b].cpu().numpy(), g["%s%d" % (k, i)])
        assert np.array_equal(meta["v_action_ids"][b].cpu().numpy(), g["vid%d" % i])
        assert np.array_equal(meta["a_action_ids"][b].cpu().numpy(), g["aid%d" % i])
    # against the oracle's collate (shapes of the absent modality included)
    ref = D.collate([D.getitem(tb, i, g["va%d" % i], g["aa%d" % i]) for i in idx])
    assert tuple(v.shape) == ref[0].shape and tuple(a.shape) == ref[1].shape and tuple(t.shape) == ref[2].shape
    assert len(meta["v_narration_ids"]) == tb["max_visual_actions"]
    if tb["max_visual_actions"]:
        assert meta["v_narration_ids"][0][1] == tb["windows"][0]["v_narration_ids"][0]      # batch row 1 is window 0
    # default draw: indices in range, data comes from the right video rows
    v2, a2, t2, _, _ = ds.batch(idx)
    torch.cuda.synchronize()
    assert torch.equal(t2, t)
    if has_v:
        w = tb["windows"][idx[0]]
        cand = tb["v_feats"][w["video_id"]][w["feat_indices"]]          # [nf, num_aug, C]
        got = v2[0].cpu().numpy()
        assert all(any(np.array_equal(got[j], cand[j, q]) for q in range(cand.shape[1])) for j in range(got.shape[0]))
