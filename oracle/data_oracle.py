"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's sample assembly,
recognition/time_interval_machine/datasets/sliding_window.py:341-421 (`__getitem__`), with the augmentation indices given
instead of drawn, plus the default collate of a list of samples.  Pinned by tests/golden/batch_*.npz, generated from the
reference's own `__getitem__` (tests/golden/make_golden_batch.py).  Never imported by tim_amd."""
import numpy as np


def getitem(ds, index, v_aug, a_aug):
    """ds: dict with windows, v_feats, v_feat_times, a_feats, a_feat_times, num_feats, window_size, max_visual_actions,
    max_audio_actions, model_modality (numpy arrays inside)."""
    w = ds["windows"][index]
    vid, fi = w["video_id"], np.asarray(w["feat_indices"])
    times = np.zeros((0, 2), np.float32)
    v_data, a_data = np.zeros((0,), np.float32), np.zeros((0,), np.float32)
    if "visual" in ds["model_modality"]:                       # :352-360
        v_data = np.asarray(ds["v_feats"][vid])[fi, np.asarray(v_aug)]
        times = np.concatenate([times, np.asarray(ds["v_feat_times"][vid], np.float32)[fi, :2]])
    if "audio" in ds["model_modality"]:                        # :362-372
        a_data = np.asarray(ds["a_feats"][vid])[fi, np.asarray(a_aug)]
        times = np.concatenate([times, np.asarray(ds["a_feat_times"][vid], np.float32)[fi, :2]])
    mv, ma = ds["max_visual_actions"], ds["max_audio_actions"]

    def pad(x, n, value, cols):                                # :374-399
        x = np.asarray(x).reshape(-1, cols) if cols else np.asarray(x).reshape(-1)
        shape = (n - x.shape[0],) + x.shape[1:]
        return np.concatenate([x, np.full(shape, value, x.dtype)])

    vq, aq = pad(np.asarray(w["v_queries"], np.float32), mv, 0.0, 2), pad(np.asarray(w["a_queries"], np.float32), ma, 0.0, 2)
    vl, al = pad(np.asarray(w["v_labels"], np.int64), mv, -1, 4), pad(np.asarray(w["a_labels"], np.int64), ma, -1, 4)
    times = np.concatenate([times, vq, aq]).astype(np.float32)  # :402-404
    times = np.maximum((times - np.float32(w["start_sec"])) / np.float32(ds["window_size"]), np.float32(0.0))
    label = {"verb": vl[:, 0], "noun": vl[:, 1], "action": vl[:, 2], "class_id": al[:, 3]}
    meta = {"v_action_ids": pad(np.asarray(w["v_action_ids"], np.int64), mv, -1, 0),
            "a_action_ids": pad(np.asarray(w["a_action_ids"], np.int64), ma, -1, 0)}
    return v_data, a_data, times, label, meta


def collate(samples):
    """torch default_collate restricted to what the sample holds"""
    stack = lambda xs: np.stack(xs) if xs[0].size else np.zeros((len(xs), 0), np.float32)
    return (stack([s[0] for s in samples]), stack([s[1] for s in samples]), np.stack([s[2] for s in samples]),
            {k: np.stack([s[3][k] for s in samples]) for k in samples[0][3]},
            {k: np.stack([s[4][k] for s in samples]) for k in samples[0][4]})
