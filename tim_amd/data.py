"""Sliding-window batch assembly on the device.

The reference's `SlidingWindowDataset` (recognition/time_interval_machine/datasets/sliding_window.py) keeps every video's
features in host memory and builds each sample in a DataLoader worker: a fancy-index gather
`feats[video][feat_indices, aug_indices]` (:358, :370), padding of the queries (:374-399) and the time normalisation
(:402-404); the batch then crosses PCIe.  At 200 k interval queries/s (2 600 windows/s x 0.67 MB) that is 1.7 GB/s of
gather + host-to-device copy per GPU.  `DeviceWindowDataset` holds the same tables in HBM (a feature set of tens of GB fits
the MI355X's 288 GB many times over) and produces whole batches with two HIP kernels (`timhip_window_gather`,
`timhip_window_times`): `batch(indices)` returns what `default_collate([dataset[i] for i in indices])` returns in the
reference, already on the device.  No CPU path: the kernels run or the call raises.
"""
import torch

from . import _lib as L
from ._lib import call, ptr
from .functional import _stream


class DeviceWindowDataset:
    def __init__(self, windows, num_feats, window_size, max_visual_actions, max_audio_actions, model_modality="audio_visual",
                 v_feats=None, v_feat_times=None, a_feats=None, a_feat_times=None, device="cuda"):
        """Arguments are the attributes the reference dataset holds after its constructor: `windows` (list of dicts,
        sliding_window.py:262-277), `v_feats`/`a_feats` {video_id: [N_feat, num_aug, C]}, `v_feat_times`/`a_feat_times`
        {video_id: [N_feat, >=2]}, `num_feats`, `window_size`, `max_visual_actions`, `max_audio_actions`."""
        if not torch.cuda.is_available():
            raise L.TimHipError("DeviceWindowDataset needs the MI355X (no CPU fallback)")
        L.load()
        self.device = torch.device(device)
        self.num_feats, self.window_size = int(num_feats), float(window_size)
        self.max_visual_actions, self.max_audio_actions = int(max_visual_actions), int(max_audio_actions)
        self.model_modality = model_modality
        self.has_v, self.has_a = "visual" in model_modality, "audio" in model_modality
        W = len(windows)
        vids = sorted({w["video_id"] for w in windows})
        self.v = self._store(v_feats, v_feat_times, vids) if self.has_v else None
        self.a = self._store(a_feats, a_feat_times, vids) if self.has_a else None
        dev, mv, ma = self.device, self.max_visual_actions, self.max_audio_actions
        fi = torch.stack([torch.as_tensor(w["feat_indices"]).to(torch.int32).reshape(-1) for w in windows])
        if fi.shape[1] != self.num_feats:
            raise ValueError("every window must index num_feats features")
        self.feat_indices = fi.to(dev).contiguous()
        self.start_sec = torch.tensor([float(w["start_sec"]) for w in windows], dtype=torch.float32, device=dev)
        vq, aq = torch.zeros((W, mv, 2)), torch.zeros((W, ma, 2))
        vl, al = torch.full((W, mv, 4), -1, dtype=torch.int64), torch.full((W, ma, 4), -1, dtype=torch.int64)
        vid_, aid_ = torch.full((W, mv), -1, dtype=torch.int64), torch.full((W, ma), -1, dtype=torch.int64)
        self.v_narration_ids, self.a_narration_ids = [], []
        for i, w in enumerate(windows):                       # the padding of sliding_window.py:374-399, done once
            T = torch.as_tensor
            nv, na = T(w["v_labels"]).shape[0], T(w["a_labels"]).shape[0]
            if nv > mv or na > ma:
                raise ValueError("window %d has more queries than max_visual_actions / max_audio_actions" % i)
            if nv:
                vq[i, :nv], vl[i, :nv], vid_[i, :nv] = T(w["v_queries"]).float(), T(w["v_labels"]).long(), T(w["v_action_ids"]).long()
            if na:
                aq[i, :na], al[i, :na], aid_[i, :na] = T(w["a_queries"]).float(), T(w["a_labels"]).long(), T(w["a_action_ids"]).long()
            self.v_narration_ids.append(list(w["v_narration_ids"]) + [""] * (mv - nv))
            self.a_narration_ids.append(list(w["a_narration_ids"]) + [""] * (ma - na))
        self.v_queries, self.a_queries = vq.to(dev), aq.to(dev)
        self.v_labels, self.a_labels = vl.to(dev), al.to(dev)
        self.v_action_ids, self.a_action_ids = vid_.to(dev), aid_.to(dev)
        row0 = lambda st: torch.tensor([st["row0"][w["video_id"]] for w in windows], dtype=torch.int64, device=dev)
        self.v_row0 = row0(self.v) if self.has_v else None
        self.a_row0 = row0(self.a) if self.has_a else None
        self.num_windows = W

    @classmethod
    def from_reference(cls, ds, device="cuda"):
        """from a constructed reference `SlidingWindowDataset` (or anything exposing the same attributes)"""
        return cls(ds.windows, ds.num_feats, ds.window_size, ds.max_visual_actions, ds.max_audio_actions, ds.model_modality,
                   ds.v_feats, ds.v_feat_times, ds.a_feats, ds.a_feat_times, device)

    def _store(self, feats, feat_times, vids):
        if feats is None or feat_times is None:
            raise ValueError("features and feature times are required for every modality of model_modality")
        row0, off, fl, tl = {}, 0, [], []
        num_aug, C = None, None
        for v in vids:
            f, t = torch.as_tensor(feats[v]), torch.as_tensor(feat_times[v]).float()
            if num_aug is None:
                num_aug, C = f.shape[1], f.shape[2]
            if f.shape[1] != num_aug or f.shape[2] != C or t.shape[0] != f.shape[0]:
                raise ValueError("inconsistent feature arrays for video %s" % v)
            row0[v] = off
            off += f.shape[0]
            fl.append(f.float().reshape(-1, C))
            tl.append(t[:, :2])
        return {"row0": row0, "num_aug": num_aug, "C": C, "feats": torch.cat(fl).to(self.device).contiguous(),
                "times": torch.cat(tl).to(self.device).contiguous()}

    def __len__(self):
        return self.num_windows

    def batch(self, indices, v_aug_indices=None, a_aug_indices=None):
        """= default_collate([dataset[i] for i in indices]) of the reference (sliding_window.py:341-421), on the device.
        aug indices [B, num_feats] pin the augmentation draw (tests); by default they are drawn on the device."""
        dev, nf = self.device, self.num_feats
        win = torch.as_tensor(indices).to(device=dev, dtype=torch.int32).contiguous().reshape(-1)
        B = win.numel()
        st = _stream()

        def gather(store, row0, aug):
            if aug is None:
                aug = torch.randint(0, store["num_aug"], (B, nf), device=dev, dtype=torch.int32)
            aug = torch.as_tensor(aug).to(device=dev, dtype=torch.int32).contiguous()
            out = torch.empty((B, nf, store["C"]), dtype=torch.float32, device=dev)
            call("timhip_window_gather", ptr(store["feats"]), store["C"], store["num_aug"], ptr(row0), ptr(self.feat_indices),
                 nf, ptr(win), B, ptr(aug), ptr(out), st)
            return out

        v_data = gather(self.v, self.v_row0, v_aug_indices) if self.has_v else torch.empty((B, 0), device=dev)
        a_data = gather(self.a, self.a_row0, a_aug_indices) if self.has_a else torch.empty((B, 0), device=dev)
        mv, ma = self.max_visual_actions, self.max_audio_actions
        T = (nf if self.has_v else 0) + (nf if self.has_a else 0) + mv + ma
        times = torch.empty((B, T, 2), dtype=torch.float32, device=dev)
        call("timhip_window_times", ptr(self.v["times"]) if self.has_v else None, 2, ptr(self.v_row0) if self.has_v else None,
             ptr(self.a["times"]) if self.has_a else None, 2, ptr(self.a_row0) if self.has_a else None,
             ptr(self.feat_indices), nf, ptr(win), B, ptr(self.v_queries), mv, ptr(self.a_queries), ma,
             ptr(self.start_sec), self.window_size, ptr(times), st)
        wl = win.long()
        vl, al = self.v_labels[wl], self.a_labels[wl]
        label = {"verb": vl[:, :, 0], "noun": vl[:, :, 1], "action": vl[:, :, 2], "class_id": al[:, :, 3]}
        idx = [int(i) for i in torch.as_tensor(indices).reshape(-1).tolist()]
        metadata = {
            "v_action_ids": self.v_action_ids[wl], "a_action_ids": self.a_action_ids[wl],
            # default_collate transposes lists of strings: one list (of B strings) per query slot
            "v_narration_ids": [[self.v_narration_ids[i][q] for i in idx] for q in range(mv)],
            "a_narration_ids": [[self.a_narration_ids[i][q] for i in idx] for q in range(ma)],
            "num_v_queries": torch.full((B,), mv, dtype=torch.int64), "num_a_queries": torch.full((B,), ma, dtype=torch.int64),
        }
        return v_data, a_data, times, label, metadata
