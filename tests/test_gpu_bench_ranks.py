"""bench.py's multi-rank control flow (launch contract: torch.distributed.run, one rank per GPU, barrier + max-over-ranks
timing, rank 0 prints ONE JSON line) exercised on a one-GPU box: two ranks share GPU 0 and the gradient all-reduce of
tim_amd/dp.py runs over gloo on the device buffers (TIM_AMD_BENCH_SHARE_GPU=1).  Catches rank-asymmetric collectives, which a
single-rank run cannot."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_one_json_line():
    env = dict(os.environ, TIM_AMD_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "40", "--no-per-shape"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 80 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 80 * 25 * 3 / (d["ms_per_step"] * 3 / 1e3)) / d["value"] < 1e-2
    # 48 NT launches + the weight gradients of the six layers as three paired launches (round 6; layer by layer: 54)
    assert d["roofline"]["launches_timed"] == 51 and d["roofline"]["achieved"] > 0
    assert "cpu_baseline" not in d          # rank 0 at N = 1 only


def test_data_parallel_step_as_graph_replay_on_one_rank():
    """bench.py's data-parallel path the way the driver's N > 1 runs take it (`--step-mode auto`): DataParallel over RCCL with the
    reduce-scatter + all-gather exchange, the whole step - collectives on the comm stream included - captured once and replayed,
    the ranks agreeing on replay-or-eager, the exchange-free comparison step captured as well.  One rank (TIM_AMD_BENCH_FORCE_DP=1:
    every collective a copy) is what a one-GPU box can run; the arithmetic of W > 1 is covered on gloo (tests/test_dp_gloo.py)."""
    env = dict(os.environ, TIM_AMD_BENCH_FORCE_DP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29573")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--batch", "40", "--no-per-shape"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["step_mode"].startswith("hip_graph_replay"), d["step_mode"]
    assert d["comm"]["collective"] == "rs_ag" and d["comm"]["without_exchange_timed_as"] == "hip_graph_replay", d["comm"]
    assert d["comm"]["comm_stream_busy_ms"] > 0 and d["comm"]["bytes_sent_plus_received_per_rank"] == 0   # one rank: nothing on a wire
    assert d["value"] > 0 and d["roofline"]["launches_timed"] == 51   # (48 NT + three paired weight-gradient launches)
    assert "cpu_baseline" not in d and "forced_one_rank_dp" in d
