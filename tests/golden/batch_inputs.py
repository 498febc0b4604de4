"""Seeded in-memory tables of a small sliding-window dataset (what the reference's constructor builds from its pickles and
.npy files), shared by make_golden_batch.py and the tests."""
import numpy as np

from tim_amd import synth


def make_tables(seed, modality, nf=6, num_aug=3, Cv=8, Ca=12):
    rs = np.random.RandomState(seed)
    vids = ["P01_%02d" % k for k in range(3)]
    has_v, has_a = "visual" in modality, "audio" in modality
    v_feats, v_ft, a_feats, a_ft = {}, {}, {}, {}
    for k, v in enumerate(vids):
        n = 40 + 17 * k
        st = (np.arange(n) * 0.2).astype(np.float32)
        ft = np.stack([st, st + 1.0, st + 0.5], 1).astype(np.float32)          # start, end, (unused third column)
        if has_v:
            v_feats[v] = synth.normal(seed, "vf" + v, (n, num_aug, Cv)).astype(np.float32); v_ft[v] = ft
        if has_a:
            a_feats[v] = synth.normal(seed, "af" + v, (n, num_aug, Ca)).astype(np.float32); a_ft[v] = ft.copy()
    max_v, max_a = (4 if has_v else 0), (3 if has_a else 0)
    windows = []
    for i in range(7):
        v = vids[i % 3]
        n = 40 + 17 * (i % 3)
        first = int(rs.randint(0, n - 2 * nf))
        fi = np.arange(first, first + 2 * nf, 2).astype(np.int64)                # feat_stride 2
        start = float(first * 0.2)
        nv, na = (int(rs.randint(0, max_v + 1)) if has_v else 0), (int(rs.randint(0, max_a + 1)) if has_a else 0)
        if i == 0:
            nv, na = max_v, max_a
        q = lambda m: (start - 0.3 + np.sort(rs.rand(m, 2) * 2.8, axis=1)).astype(np.float32)   # some start before the window
        lab = lambda m, audio: np.stack([rs.randint(0, 97, m), rs.randint(0, 300, m), rs.randint(0, 3806, m),
                                         rs.randint(0, 44, m) if audio else np.full(m, -1)], 1).astype(np.int64).reshape(m, 4)
        windows.append({"video_id": v, "start_sec": start, "stop_sec": start + nf * 0.4, "feat_indices": fi,
                        "v_queries": q(nv), "v_labels": lab(nv, False), "v_action_ids": rs.randint(0, 9999, nv).astype(np.int64),
                        "v_narration_ids": ["v_%d_%d" % (i, j) for j in range(nv)],
                        "a_queries": q(na), "a_labels": lab(na, True), "a_action_ids": rs.randint(0, 9999, na).astype(np.int64),
                        "a_narration_ids": ["a_%d_%d" % (i, j) for j in range(na)]})
    return {"windows": windows, "v_feats": v_feats if has_v else None, "v_feat_times": v_ft if has_v else None,
            "a_feats": a_feats if has_a else None, "a_feat_times": a_ft if has_a else None, "num_feats": nf,
            "window_size": nf * 0.2 * 2, "max_visual_actions": max_v, "max_audio_actions": max_a, "num_aug": num_aug,
            "model_modality": modality}
