"""Seeded inputs shared by make_golden_nms.py (generation, build container) and the NMS tests (regeneration)."""
import numpy as np

from tim_amd import synth


def make_segments(seed, n, ties):
    """segments in [0, 100) s with lengths 0.2 .. 20 s; scores in (0, 1); `ties`: quantise scores so that equal values occur"""
    u = synth.uniform01(seed, "nms", (n, 3))
    start = (u[:, 0] * 100.0).astype(np.float32)
    length = (0.2 + u[:, 1] * u[:, 1] * 19.8).astype(np.float32)
    segs = np.stack([start, start + length], 1).astype(np.float32)
    scores = u[:, 2].astype(np.float32)
    if ties:
        scores = (np.floor(scores * 16) / 16 + 1 / 32).astype(np.float32)
    return segs, scores


def make_classes(seed, n, ncls):
    return np.floor(synth.uniform01(seed, "nms_cls", (n,)) * ncls).astype(np.int64)
