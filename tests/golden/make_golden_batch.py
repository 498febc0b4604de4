"""Golden vectors for the sliding-window batch assembly from the reference's own `__getitem__`
(build container only):    python tests/golden/make_golden_batch.py

A reference `SlidingWindowDataset` object is created WITHOUT its file-reading constructor (object.__new__) and given the
in-memory tables its constructor would have built, from seeded synthetic data (tests/golden/batch_inputs.py); then the
reference's own `__getitem__` (datasets/sliding_window.py:341-421) runs for every window.  Its augmentation draw
(torch.randint, :352-357, :364-369) is replayed with the same seed and stored with the outputs.
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
from time_interval_machine.datasets.sliding_window import SlidingWindowDataset  # noqa: E402

from tests.golden.batch_inputs import make_tables  # noqa: E402


def run(name, seed, modality):
    tb = make_tables(seed, modality)
    ds = object.__new__(SlidingWindowDataset)
    t = lambda d: None if d is None else {k: torch.from_numpy(v) for k, v in d.items()}
    ds.windows = [{k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in w.items()} for w in tb["windows"]]
    ds.v_feats, ds.v_feat_times, ds.a_feats, ds.a_feat_times = t(tb["v_feats"]), t(tb["v_feat_times"]), t(tb["a_feats"]), t(tb["a_feat_times"])
    ds.model_modality, ds.num_feats, ds.window_size = modality, tb["num_feats"], tb["window_size"]
    ds.max_visual_actions, ds.max_audio_actions = tb["max_visual_actions"], tb["max_audio_actions"]
    ds.v_num_aug, ds.a_num_aug = tb["num_aug"], tb["num_aug"]
    out = {}
    for i in range(len(ds.windows)):
        torch.manual_seed(1000 + i)
        v, a, times, label, meta = ds[i]
        torch.manual_seed(1000 + i)                                    # replay the draws of :352-357 / :364-369
        va = torch.randint(low=0, high=ds.v_num_aug, size=(ds.num_feats,), dtype=torch.long) if "visual" in modality else torch.zeros(0)
        aa = torch.randint(low=0, high=ds.a_num_aug, size=(ds.num_feats,), dtype=torch.long) if "audio" in modality else torch.zeros(0)
        out.update({"v%d" % i: v.numpy(), "a%d" % i: a.numpy(), "t%d" % i: times.numpy(), "va%d" % i: va.numpy(), "aa%d" % i: aa.numpy(),
                    "verb%d" % i: label["verb"].numpy(), "noun%d" % i: label["noun"].numpy(), "action%d" % i: label["action"].numpy(),
                    "class_id%d" % i: label["class_id"].numpy(), "vid%d" % i: meta["v_action_ids"].numpy(),
                    "aid%d" % i: meta["a_action_ids"].numpy()})
        assert meta["num_v_queries"] == tb["max_visual_actions"] and len(meta["v_narration_ids"]) == tb["max_visual_actions"]
    np.savez(os.path.join(HERE, "batch_%s.npz" % name), seed=seed, modality=modality, n=len(ds.windows), **out)
    print(name, "windows", len(ds.windows))


if __name__ == "__main__":
    run("av", 71, "audio_visual")
    run("visual", 72, "visual")
    run("audio", 73, "audio")
