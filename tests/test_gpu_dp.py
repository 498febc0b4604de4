"""Data-parallel wrapper on a real GPU: a 1-rank RCCL group, with the wrapper told world=2, drives every gradient
bucket through the comm-stream all-reduce (identity on one rank) and the 1/world scaling; gradients must be exactly
half of a plain run with the same dropout seed.  (The 2-rank arithmetic itself is covered on CPU/gloo.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dp_bucket_allreduce_path_on_one_gpu():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_single_gpu_check.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert "RESULT params 102 mismatches 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
