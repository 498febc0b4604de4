"""1-D segment NMS: oracle/nms_oracle.c against vectors produced by the reference's own nms_cpu.cpp and
nms.py (tests/golden/make_golden_nms.py).  CPU only; bit-exact."""
import glob
import os

import numpy as np
import pytest

from oracle import nms_oracle as N
from tests.helpers import GOLDEN
from tests.golden.make_golden_nms_inputs import make_segments, make_classes

CASES = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "nms_*.npz")) if "batched" not in f)
BATCHED = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "nms_batched_*.npz")))


@pytest.mark.parametrize("case", CASES)
def test_nms_oracle_matches_reference(case):
    g = np.load(os.path.join(GOLDEN, case))
    segs, scores = make_segments(int(g["seed"]), int(g["n"]), bool(g["ties"]))
    keep = N.nms_1d(segs, scores, float(g["iou"]), order=g["order"])
    assert np.array_equal(keep, g["keep"])
    inds, dets = N.softnms_1d(segs, scores, float(g["iou"]), float(g["sigma"]), float(g["min_score"]), int(g["method"]))
    assert np.array_equal(inds, g["soft_inds"])
    assert np.array_equal(dets, g["soft_dets"])           # bit-exact scores too (same libm expf on the same host)


@pytest.mark.parametrize("case", BATCHED)
def test_batched_nms_oracle_matches_reference(case):
    g = np.load(os.path.join(GOLDEN, case))
    segs, scores = make_segments(int(g["seed"]), int(g["n"]), False)
    cls = make_classes(int(g["seed"]), int(g["n"]), int(g["ncls"]))
    assert np.array_equal(cls, g["cls"])
    S, Cc, Ll = N.batched_nms(segs, scores, cls, 0.1, 0.001, sigma=0.4, method=2, nms=str(g["kind"]))
    assert np.array_equal(Cc, g["scores"]) and np.array_equal(S, g["segs"]) and np.array_equal(Ll, g["labels"])
