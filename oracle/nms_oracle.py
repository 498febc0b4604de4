"""TEST INFRASTRUCTURE ONLY: ctypes front of oracle/nms_oracle.c (the CPU restatement of the reference's
detection/eval_detection/csrc/nms_cpu.cpp) plus a numpy restatement of eval_detection/nms.py:batched_nms on top of it.
Built by `make -C oracle` (also from __graft_entry__.build()).  Never imported by tim_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libnms_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE])
        _lib = C.CDLL(so)
        _lib.nms_1d_oracle.restype = C.c_int64
        _lib.nms_1d_oracle.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]
        _lib.softnms_1d_oracle.restype = C.c_int64
        _lib.softnms_1d_oracle.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int,
                                           C.c_void_p, C.c_void_p]
    return _lib


def nms_1d(segs, scores, iou_threshold, order=None):
    """nms_cpu.cpp:19-60 -> kept original indices in descending-score order.  `order` defaults to a stable descending
    argsort (the reference's torch sort leaves the order of equal scores unspecified)."""
    segs = np.ascontiguousarray(segs, dtype=np.float32)
    n = segs.shape[0]
    if order is None:
        order = np.argsort(-np.asarray(scores, dtype=np.float32), kind="stable")
    order = np.ascontiguousarray(order, dtype=np.int64)
    keep = np.empty(n, dtype=np.int64)
    m = _load().nms_1d_oracle(segs.ctypes.data, order.ctypes.data, n, float(iou_threshold), keep.ctypes.data)
    return keep[:m]


def softnms_1d(segs, scores, iou_threshold, sigma, min_score, method):
    """nms_cpu.cpp:69-170 -> (inds [m], dets [m,3])"""
    segs = np.ascontiguousarray(segs, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    n = segs.shape[0]
    dets = np.zeros((n, 3), dtype=np.float32)
    inds = np.empty(n, dtype=np.int64)
    m = _load().softnms_1d_oracle(segs.ctypes.data, scores.ctypes.data, n, float(iou_threshold), float(sigma),
                                  float(min_score), int(method), dets.ctypes.data, inds.ctypes.data)
    return inds[:m], dets[:m]


def batched_nms(segs, scores, cls_idxs, iou_threshold, min_score, sigma=0.5, method=2, nms="soft", max_seg_num=2000000):
    """eval_detection/nms.py:97-180, multi_class=True branch: per class (ascending class id, torch.unique order) NMS, concatenate,
    final descending sort by score (stable here)."""
    segs, scores, cls_idxs = np.asarray(segs, np.float32), np.asarray(scores, np.float32), np.asarray(cls_idxs)
    if segs.shape[0] == 0:
        return np.zeros((0, 2), np.float32), np.zeros((0,), np.float32), np.zeros((0,), cls_idxs.dtype)
    out_s, out_c, out_l = [], [], []
    for cid in np.unique(cls_idxs):
        cur = np.where(cls_idxs == cid)[0]
        if nms == "soft":
            inds, dets = softnms_1d(segs[cur], scores[cur], iou_threshold, sigma, min_score, method)
            out_s.append(dets[:, :2]); out_c.append(dets[:, 2]); out_l.append(cls_idxs[cur][inds])
        else:
            s, c, l = segs[cur], scores[cur], cls_idxs[cur]
            if min_score > 0:                                   # nms.py:16-20
                m = c > min_score
                s, c, l = s[m], c[m], l[m]
            keep = nms_1d(s, c, iou_threshold)
            if max_seg_num > 0:
                keep = keep[:max_seg_num]
            out_s.append(s[keep]); out_c.append(c[keep]); out_l.append(l[keep])
    S, Cc, Ll = np.concatenate(out_s), np.concatenate(out_c), np.concatenate(out_l)
    idx = np.argsort(-Cc, kind="stable")
    return S[idx], Cc[idx], Ll[idx]
