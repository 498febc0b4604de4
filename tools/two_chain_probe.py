#!/usr/bin/env python3
"""Round-5 structural probe: does the C2a step get shorter when the batch runs as TWO half-batch chains side by side?

The step is a serial sum of MFMA-bound main loops and fabric-bound epilogues / LayerNorms / attention.
A GEMM block owns its CU (156 KiB of LDS, 504 of 512 VGPRs per SIMD lane), so nothing co-runs ON a CU - but a half-batch GEMM
has 124 / 248 / 372 tiles, i.e. leaves half of the CUs to the other chain, and an HBM-bound kernel barely slows down on half of
the CUs (tools/cumask_probe.py: LayerNorm-backward 31 -> 35 us on 128 CUs).  If chain B runs half a layer behind chain A, A's
epilogue bursts and row kernels fall under B's main loops.

Arms (each a HIP-graph replay, median of per-replay HIP events):
  full      one model, B windows, one stream                                   (the bench line's step)
  seq       two models, B/2 windows each, one stream, back to back             (what halving the batch costs by itself)
  par(d)    two models, B/2 each, two streams forked / joined inside the graph; chain B starts after chain A's first `d`
            launches... realised as: chain B waits for an event recorded on chain A after its time MLP + `d` encoder layers of
            a SEPARATE warm-up model (a delay of d layer-forwards of GPU time)
The two half-batch models have their own weights (same shapes): only the kernels' timing matters here.

    TIMHIP_GEMM_PP_MIN_TILES=96 python tools/two_chain_probe.py [--batch 64] [--delays 0,1,2]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from tim_amd import functional as F  # noqa: E402
from tim_amd.config import named_config  # noqa: E402


def capture(fn, dev, warmup=3):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(warmup):
            fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def time_graph(g, reps=40, warm=10):
    for _ in range(warm):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--delays", default="0,1,2,3")
    ap.add_argument("--spin-us", default="0,30,60,100,150", help="delay of chain B by a spin kernel of this many microseconds")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg, nv, na = named_config("C2a"), 15, 10
    B = a.batch
    F.graph_safe_dropout(dev)
    res = {"batch": B, "precision": a.precision, "pp_min_tiles": os.environ.get("TIMHIP_GEMM_PP_MIN_TILES", "192")}

    def mk(bsz, seed):
        m, _ = bench.build_model(cfg, a.precision, dev, seed=seed)
        m.train()
        m._ws_pinned = True
        return m, bench.make_batch(cfg, bsz, nv, na, seed + 5, dev), [None]

    mf, bf, Rf = mk(B, 0)
    g_full = capture(lambda: bench.step_fn(mf, bf, nv, na, Rf), dev)
    res["full_ms"], res["full_min_ms"] = time_graph(g_full)
    print("full: %.3f ms" % res["full_ms"], flush=True)
    del g_full

    ma, ba, Ra = mk(B // 2, 1)
    mb, bb, Rb = mk(B // 2, 2)

    def seq():
        bench.step_fn(ma, ba, nv, na, Ra)
        bench.step_fn(mb, bb, nv, na, Rb)
    g_seq = capture(seq, dev)
    res["seq_ms"], res["seq_min_ms"] = time_graph(g_seq)
    print("two half-batch steps back to back: %.3f ms" % res["seq_ms"], flush=True)
    del g_seq

    s2 = torch.cuda.Stream(device=dev)
    res["par"] = {}
    for us in [int(x) for x in a.spin_us.split(",")]:
        def par():
            main_s = torch.cuda.current_stream(dev)
            s2.wait_stream(main_s)
            with torch.cuda.stream(s2):
                if us > 0:
                    torch.cuda._sleep(int(us * 2100))   # ~2.1 GHz shader clock: a delay, not a measurement
                bench.step_fn(mb, bb, nv, na, Rb)
            bench.step_fn(ma, ba, nv, na, Ra)
            main_s.wait_stream(s2)
        g = capture(par, dev)
        med, mn = time_graph(g)
        res["par"]["spin_%dus" % us] = {"ms": med, "min_ms": mn}
        print("two chains side by side, chain B delayed %d us: %.3f ms (min %.3f)" % (us, med, mn), flush=True)
        del g
    print(json.dumps(res))


if __name__ == "__main__":
    main()
