"""Which RCCL call patterns survive HIP-graph capture on this stack (1-rank group, one GPU)?  Each case runs in its own
process (a failure here is a segmentation fault inside hipStreamEndCapture / hipGraphInstantiate, not an exception).

    python tools/rccl_capture_probe.py            # runs every case, prints one line per case
    python tools/rccl_capture_probe.py --case a2a # one case in this process
"""
import argparse
import os
import subprocess
import sys

CASES = ["allreduce", "allgather", "reduce_scatter", "allreduce_side_stream", "rs_ag_side_stream", "copy_side_stream", "a2a", "a2a_side_stream"]


def run_case(case):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29519")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    x = torch.arange(1 << 16, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    comm = torch.cuda.Stream(device=dev)

    def body():
        if case == "allreduce":
            dist.all_reduce(x)
        elif case == "a2a":
            dist.all_to_all_single(y, x)
        elif case == "allgather":
            dist.all_gather_into_tensor(y, x)
        elif case == "reduce_scatter":
            dist.reduce_scatter_tensor(y, x)
        elif case in ("a2a_side_stream", "allreduce_side_stream", "copy_side_stream", "rs_ag_side_stream"):
            comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(comm):
                if case == "a2a_side_stream":
                    dist.all_to_all_single(y, x)
                    dist.all_gather_into_tensor(x, y)
                elif case == "allreduce_side_stream":
                    dist.all_reduce(x)
                elif case == "rs_ag_side_stream":
                    dist.reduce_scatter_tensor(y, x)
                    dist.all_gather_into_tensor(x, y)
                else:
                    y.copy_(x)
            torch.cuda.current_stream().wait_stream(comm)
        elif case == "dp_reduce_only":
            y.copy_(x).mul_(0.5)
        x.add_(1.0)

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    torch.cuda.synchronize()
    before = x.clone()
    g.replay()
    torch.cuda.synchronize()
    print("CASE %s ok: replay advanced x by %.1f" % (case, (x - before).abs().max().item()), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default=None)
    a = ap.parse_args()
    if a.case:
        run_case(a.case)
        return
    for i, c in enumerate(CASES):
        env = dict(os.environ, MASTER_PORT=str(29540 + i))
        try:
            out = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), "--case", c], env=env,
                                 capture_output=True, text=True, timeout=60)
        except subprocess.TimeoutExpired:
            print("CASE %s HUNG (no result within 60 s; the process was killed)" % c, flush=True)
            continue
        ok = [ln for ln in out.stdout.splitlines() if ln.startswith("CASE")]
        if ok:
            print(ok[0])
        else:
            where = [ln.strip() for ln in out.stderr.splitlines() if "File" in ln or "Error" in ln or "error" in ln][:4]
            print("CASE %s FAILED rc=%d: %s" % (c, out.returncode, " | ".join(where)[:500]))


if __name__ == "__main__":
    main()
