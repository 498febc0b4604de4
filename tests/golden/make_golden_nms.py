"""Golden vectors for the 1-D segment NMS from the reference's own C++ (build container only).

    python tests/golden/make_golden_nms.py

Compiles /root/reference/detection/eval_detection/csrc/nms_cpu.cpp where it lies (torch.utils.cpp_extension, build
directory oracle/_ref/, git-ignored) and runs nms / softnms on seeded inputs; stores inputs' seeds and the outputs
(kept indices, dets) as plain numbers in tests/golden/nms_*.npz.  Also runs the reference's eval_detection/nms.py
batched_nms on top of it for two end-to-end cases.
"""
import os
import sys

import numpy as np
import torch
from torch.utils.cpp_extension import load

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tim_amd import synth  # noqa: E402

REF = "/root/reference/detection/eval_detection"
bdir = os.path.join(ROOT, "oracle", "_ref")
os.makedirs(bdir, exist_ok=True)
ext = load(name="nms_1d_cpu", sources=[os.path.join(REF, "csrc", "nms_cpu.cpp")], build_directory=bdir,
           extra_cflags=["-O2", "-fopenmp"], verbose=False)
sys.modules["nms_1d_cpu"] = ext
sys.path.insert(0, REF)
import nms as ref_nms  # noqa: E402


from tests.golden.make_golden_nms_inputs import make_segments, make_classes  # noqa: E402


def case(name, seed, n, ties, iou, sigma, min_score, method):
    segs, scores = make_segments(seed, n, ties)
    s, c = torch.from_numpy(segs), torch.from_numpy(scores)
    keep = ext.nms(s.clone(), c.clone(), iou_threshold=float(iou))
    order = torch.sort(c, 0, descending=True)[1]          # the order nms_cpu.cpp:28 used (same torch, same call)
    dets = torch.zeros((n, 3), dtype=torch.float32)
    inds = ext.softnms(s.clone(), c.clone(), dets, iou_threshold=float(iou), sigma=float(sigma), min_score=float(min_score),
                       method=int(method))
    np.savez(os.path.join(HERE, "nms_%s.npz" % name), seed=seed, n=n, ties=ties, iou=iou, sigma=sigma, min_score=min_score,
             method=method, order=order.numpy(), keep=keep.numpy(), soft_inds=inds.numpy(), soft_dets=dets[:len(inds)].numpy())
    print(name, "n", n, "vanilla kept", len(keep), "soft kept", len(inds))


def batched_case(name, seed, n, ncls, nms_kind):
    segs, scores = make_segments(seed, n, False)
    cls = torch.from_numpy(make_classes(seed, n, ncls))
    o = ref_nms.batched_nms(torch.from_numpy(segs), torch.from_numpy(scores), cls, iou_threshold=0.1, min_score=0.001,
                            sigma=0.4, method=2, nms=nms_kind, multi_class=True, voting_thresh=0.75)
    np.savez(os.path.join(HERE, "nms_batched_%s.npz" % name), seed=seed, n=n, ncls=ncls, kind=nms_kind, cls=cls.numpy(),
             segs=o[0], scores=o[1], labels=o[2])
    print(name, "batched_nms kept", len(o[1]))


if __name__ == "__main__":
    case("small", 51, 40, False, 0.5, 0.5, 0.001, 2)
    case("ties", 52, 300, True, 0.3, 0.4, 0.05, 2)
    case("linear", 53, 257, False, 0.1, 0.4, 0.01, 1)
    case("vanilla", 54, 500, True, 0.1, 0.4, 0.001, 0)
    case("big", 55, 3000, False, 0.1, 0.4, 0.001, 2)
    case("one", 56, 1, False, 0.1, 0.4, 0.001, 2)
    batched_case("soft", 57, 600, 7, "soft")
    batched_case("vanilla", 58, 400, 5, "vanilla")
