"""Data-parallel wrapper on a real GPU: a 1-rank RCCL group with the exchange forced on drives every gradient bucket through
the comm-stream narrow / all-to-all / fp32-sum / all-gather / widen sequence (copies on one rank) while the backward runs;
the gradients must be the bf16 rounding of a plain run's with the same dropout seed, and the cost of the side-stream work to
the data chain is printed.  (The 2-rank arithmetic itself is covered on CPU/gloo and by the two-process test below.)"""
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
    assert "INTERFERENCE" in out.stdout
    print([ln for ln in out.stdout.splitlines() if ln.startswith("INTERFERENCE")][0])


def test_two_ranks_sharing_one_gpu_average_their_gradients():
    """two real processes, each with its half of the windows, gradient buckets reduced over gloo on the device buffers
    through tim_amd/dp.py's hooks and streams: result = half the gradients of one process run on all the windows"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29577", os.path.join(ROOT, "tools", "dp_two_rank_check.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert "RESULT params 102 mismatches 0 ranks_agree True" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert "RANK1 ranks_agree True" in out.stdout
    # gradient accumulation (first microbatch under no_sync) gives the one-pass averaged gradients on both ranks
    assert "ACCUM rank 0 mismatches 0" in out.stdout and "ACCUM rank 1 mismatches 0" in out.stdout, out.stdout[-2000:]
