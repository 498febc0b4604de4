"""The end-to-end example (device-side batches, encoder, fused losses, AdamW) runs and learns on the synthetic data."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_train_synthetic_example_learns():
    import train_synthetic
    hist = train_synthetic.main(["--config", "tiny", "--steps", "40", "--batch", "16", "--precision", "bf16", "--lr", "3e-3"])
    assert all(h == h for h in hist)                     # finite
    assert sum(hist[-5:]) / 5 < sum(hist[:5]) / 5 - 0.05, (hist[:5], hist[-5:])


def test_train_detection_example_learns():
    import train_detection_synthetic
    hist = train_detection_synthetic.main(["--steps", "30", "--batch", "4", "--precision", "bf16"])
    assert all(h == h for h in hist)
    assert sum(hist[-5:]) / 5 < 0.8 * sum(hist[:5]) / 5, (hist[:5], hist[-5:])
