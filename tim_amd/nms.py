"""1-D segment NMS on the MI355X: host mirror of the reference's detection/eval_detection/nms.py over the
batched HIP kernels of tim_amd/csrc/nms.hip (C ABI timhip_softnms_1d / timhip_nms_1d).

`batched_nms` keeps the reference's name, arguments and return value (three numpy arrays); `grouped_nms` is the form the
GPU wants: ONE call for all (video, class) groups of an evaluation instead of one joblib task per video
(format_predictions_epic.py:146-156).  There is no CPU path: the kernels run or the call raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from ._lib import call, ptr
from .functional import _stream


def _dev(t, dtype):
    if not torch.cuda.is_available():
        raise L.TimHipError("tim_amd.nms runs on the MI355X HIP kernels only (no CPU fallback)")
    L.load()
    return torch.as_tensor(t).to(device="cuda", dtype=dtype).contiguous()


def grouped_nms(segs, scores, keys, iou_threshold, min_score, sigma=0.5, method=2, nms="soft", max_seg_num=2000000):
    """NMS applied independently to every group of equal `keys` (int64, e.g. video_index * num_classes + class).
    Returns (segs [M,2], scores [M], keys [M]) as device tensors, groups in ascending key order and, inside a group, in
    selection order (= descending score for vanilla NMS)."""
    segs, scores, keys = _dev(segs, torch.float32).reshape(-1, 2), _dev(scores, torch.float32), _dev(keys, torch.int64)
    if nms != "soft" and min_score > 0:                      # nms.py:16-20: vanilla NMS filters by score first
        m = scores > min_score
        segs, scores, keys = segs[m], scores[m], keys[m]
    n = segs.shape[0]
    if n == 0:
        return segs.new_zeros((0, 2)), scores.new_zeros((0,)), keys.new_zeros((0,))
    order = torch.argsort(keys, stable=True)
    segs, scores, keys = segs[order].contiguous(), scores[order].contiguous(), keys[order]
    ukeys, counts = torch.unique_consecutive(keys, return_counts=True)
    G = ukeys.numel()
    off = torch.zeros(G + 1, dtype=torch.int32, device=segs.device)
    off[1:] = torch.cumsum(counts, 0).to(torch.int32)
    off_host = off.cpu()
    cnt = torch.empty(G, dtype=torch.int32, device=segs.device)
    st = _stream()
    if nms == "soft":
        dets = torch.empty((n, 3), dtype=torch.float32, device=segs.device)
        inds = torch.empty(n, dtype=torch.int32, device=segs.device)
        wsb = L.load().timhip_softnms_1d_workspace_bytes(n, G)
        ws = torch.empty(wsb, dtype=torch.uint8, device=segs.device)
        call("timhip_softnms_1d", ptr(segs), ptr(scores), ptr(off), off_host.data_ptr(), G, float(iou_threshold),
             float(sigma), float(min_score), int(method), ptr(dets), ptr(inds), ptr(cnt), ptr(ws), wsb, st)
        rows = _kept_rows(off, cnt, n)
        d = dets[rows]
        return d[:, :2].contiguous(), d[:, 2].contiguous(), keys[rows]
    # vanilla: per-group descending order of the scores (stable), relative to the group's first row
    gid = torch.repeat_interleave(torch.arange(G, device=segs.device), counts)
    o = torch.argsort(-scores, stable=True)
    o = o[torch.argsort(gid[o], stable=True)]                # grouped, descending inside each group
    rel = (o - off[:-1].long()[gid[o]]).to(torch.int32).contiguous()
    keep = torch.empty(n, dtype=torch.int32, device=segs.device)
    removed = torch.empty(n, dtype=torch.uint8, device=segs.device)
    call("timhip_nms_1d", ptr(segs), ptr(rel), ptr(off), G, float(iou_threshold), ptr(removed), ptr(keep), ptr(cnt), st)
    if max_seg_num > 0:
        cnt = torch.clamp(cnt, max=int(max_seg_num))
    rows = _kept_rows(off, cnt, n)
    src = keep[rows].long() + off[:-1].long()[gid[rows]]
    return segs[src], scores[src], keys[src]


def _kept_rows(off, cnt, n):
    """row numbers off[g] .. off[g] + cnt[g] - 1 for every group, concatenated"""
    pos = torch.arange(n, device=off.device)
    gid = torch.searchsorted(off[1:].long(), pos, right=True)
    return pos[(pos - off[:-1].long()[gid]) < cnt.long()[gid]]


def seg_voting(nms_segs, all_segs, all_scores, iou_threshold, score_offset=1.5):
    """eval_detection/nms.py:61-94 (torch ops on whichever device the inputs live)"""
    num_nms_segs, num_all_segs = nms_segs.shape[0], all_segs.shape[0]
    ex_nms_segs = nms_segs[:, None].expand(num_nms_segs, num_all_segs, 2)
    ex_all_segs = all_segs[None, :].expand(num_nms_segs, num_all_segs, 2)
    left = torch.maximum(ex_nms_segs[:, :, 0], ex_all_segs[:, :, 0])
    right = torch.minimum(ex_nms_segs[:, :, 1], ex_all_segs[:, :, 1])
    inter = (right - left).clamp(min=0)
    nms_seg_lens = ex_nms_segs[:, :, 1] - ex_nms_segs[:, :, 0]
    all_seg_lens = ex_all_segs[:, :, 1] - ex_all_segs[:, :, 0]
    iou = inter / (nms_seg_lens + all_seg_lens - inter)
    seg_weights = (iou >= iou_threshold).to(all_scores.dtype) * all_scores[None, :] * iou
    seg_weights /= torch.sum(seg_weights, dim=1, keepdim=True)
    return seg_weights @ all_segs


def batched_nms(segs, scores, cls_idxs, iou_threshold, min_score, sigma=0.5, method=2, nms="soft", multi_class=True,
                voting_thresh=0.75, max_seg_num=2000000):
    """eval_detection/nms.py:97-180, same arguments; returns (segs, scores, cls_idxs) as numpy arrays sorted by descending
    score (ties keep the per-class concatenation order: the reference's unstable torch sort leaves them unspecified)."""
    segs_t, scores_t = torch.as_tensor(segs, dtype=torch.float32), torch.as_tensor(scores, dtype=torch.float32)
    cls_t = torch.as_tensor(cls_idxs)
    if segs_t.shape[0] == 0:
        return np.zeros((0, 2), np.float32), np.zeros((0,), np.float32), np.zeros((0,), cls_t.numpy().dtype)
    keys = cls_t.to(torch.int64) if multi_class else torch.zeros_like(cls_t, dtype=torch.int64)
    s, c, k = grouped_nms(segs_t, scores_t, keys, iou_threshold, min_score, sigma, method, nms, max_seg_num)
    if not multi_class:
        raise NotImplementedError("class-agnostic batched_nms (the reference's own call passes 8 arguments to a "
                                  "7-argument SoftNMSop there, nms.py:148-151) is not used by the evaluation scripts")
    idx = torch.argsort(-c, stable=True)
    return s[idx].cpu().numpy(), c[idx].cpu().numpy(), k[idx].cpu().numpy().astype(cls_t.numpy().dtype)
