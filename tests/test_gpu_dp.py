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


def test_dp_step_captured_with_its_collectives():
    """the data-parallel step - bucket exchanges (reduce-scatter + all-gather over RCCL) on the comm stream included - captured by
    GraphedStep on a one-rank RCCL group: a replay's gradients equal the eager data-parallel step's under the same dropout salt,
    and the next replay draws fresh masks (tools/dp_graph_check.py --quick)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29535")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_graph_check.py"), "--quick"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert "REPLAY params 54 mismatches 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert "REPLAY fresh_masks True" in out.stdout


def test_graphed_step_refuses_the_all_to_all_wrapper():
    """all_to_all_single under capture hangs or crashes hipStreamEndCapture on this stack (profiles/r05_rccl_capture_probe.txt):
    GraphedStep must raise BEFORE capturing, not segfault"""
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "import bench\n"
        "from tim_amd.config import named_config\n"
        "from tim_amd.dp import DataParallel\n"
        "from tim_amd.graph import GraphedStep\n"
        "m, _ = bench.build_model(named_config('tiny'), 'fp16', torch.device('cuda', 0)); m.train()\n"
        "dp = DataParallel(m, force=True, collective='a2a')\n"
        "assert dp.collective == 'a2a'\n"
        "try:\n"
        "    GraphedStep(dp, lambda: None)\n"
        "    print('NOT REFUSED')\n"
        "except RuntimeError as e:\n"
        "    print('REFUSED', 'all_to_all_single' in str(e))\n"
        "dist.destroy_process_group()\n" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "REFUSED True" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
