"""The data-parallel step as a HIP graph, on what one GPU can show (round 5; VERDICT r4 "next" #3).

A 1-rank RCCL group with the exchange forced on (`DataParallel(force=True)`): every gradient bucket goes through the comm-stream
reduce-scatter -> all-gather sequence (copies on one rank).  This tool
  (a) captures that step - RCCL collectives on the comm stream included - with `tim_amd.graph.GraphedStep` and checks that a
      replay's gradients equal the eager data-parallel step's (same dropout salt);
  (b) measures what the wrapper costs on one GPU before a byte crosses a link, eager and replayed, for
      buckets_per_exchange = 1 / 2 / 4 / all and both wire formats (profiles/r05_dp_single_gpu_overhead.txt).
If RCCL refuses the capture the error text is printed (CAPTURE_ERROR ...) and the eager figures stand alone.

    python tools/dp_graph_check.py [--batch 64] [--precision fp16] [--quick]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29514")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--quick", action="store_true", help="parity only, small batch (the test suite's form)")
    ap.add_argument("--short", action="store_true", help="rs_ag / rs_ag with stubbed collectives / allreduce only (no a2a rows)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import bench
    from tim_amd import functional as F
    from tim_amd.config import named_config
    from tim_amd.dp import DataParallel
    from tim_amd.graph import GraphedStep

    cfg = named_config("C2a") if not a.quick else named_config("tiny")
    nv, na = (15, 10) if not a.quick else (4, 2)
    B = a.batch if not a.quick else 4
    salt = F.graph_safe_dropout(dev)

    def fresh(seed=0):
        m, _ = bench.build_model(cfg, a.precision, dev, seed=seed)
        m.train()
        return m

    batch = bench.make_batch(cfg, B, nv, na, 100, dev)

    # ---- (a) replay == eager for the data-parallel step
    model = fresh()
    dp = DataParallel(model, force=True, buckets_per_exchange=2)
    assert dp.active and dp.world == 1 and dp.collective == "rs_ag", (dp.active, dp.world, dp.collective, dp._why)
    R = [None]
    fn = lambda: bench.step_fn(dp, batch, nv, na, R)   # noqa: E731
    fn()
    torch.cuda.synchronize()
    captured = None
    try:
        gs = GraphedStep(dp, fn)
        captured = gs
    except Exception as e:  # noqa: BLE001
        print("CAPTURE_ERROR %s: %s" % (type(e).__name__, str(e).replace("\n", " | ")[:600]), flush=True)
    if captured is not None:
        salt.fill_(12345)
        fn()                                  # eager DP step from salt 12345 (the step advances the salt itself)
        torch.cuda.synchronize()
        ge = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        salt.fill_(12345)
        captured()
        torch.cuda.synchronize()
        bad = 0
        for n, p in model.named_parameters():
            if p.grad is None:
                continue
            s = ge[n].abs().max().item() + 1e-12
            d = (p.grad - ge[n]).abs().max().item()
            if not d <= 1e-5 * s + 1e-7:       # fp32-atomics tolerance (tests/test_gpu_graph.py)
                bad += 1
                print("MISMATCH", n, d, s)
        print("REPLAY params %d mismatches %d" % (len(ge), bad), flush=True)
        # a second replay draws different masks: its gradients differ from the first's
        g1 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        captured()
        torch.cuda.synchronize()
        moved = sum(1 for n, p in model.named_parameters() if p.grad is not None and not torch.equal(p.grad, g1[n]))
        print("REPLAY fresh_masks %s" % (moved > 0), flush=True)
        del gs, captured
    if a.quick:
        dist.destroy_process_group()
        return

    # ---- (b) what the wrapper costs on one GPU
    def median_ms(call, n=30, warm=30):
        for _ in range(warm):
            call()
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        t0 = time.perf_counter()
        for i in range(n):
            evs[i].record()
            call()
        evs[n].record()
        issue = (time.perf_counter() - t0) / n * 1e3
        torch.cuda.synchronize()
        d = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
        return d[len(d) // 2], issue

    plain = fresh()
    Rp = [None]
    fnp = lambda: bench.step_fn(plain, batch, nv, na, Rp)   # noqa: E731
    e_plain, i_plain = median_ms(fnp)
    gp = GraphedStep(plain, fnp)
    g_plain, _ = median_ms(gp, warm=10)
    print("PLAIN   eager %.3f ms (host issue %.3f)   replay %.3f ms" % (e_plain, i_plain, g_plain), flush=True)
    del gp
    # round 6: "stub" rows - the collectives replaced by a one-element kernel (tim_amd/dp.py: _stub): the wrapper's joins, events,
    # per-layer hooks and launches WITHOUT the copies a one-rank collective amounts to
    combos = [("rs_ag", torch.float32, False), ("rs_ag", torch.float32, True), ("allreduce", torch.float32, False)]
    if not a.short:
        combos += [("a2a", torch.float32, False), ("a2a", torch.bfloat16, False)]
    for coll, wire, stub in combos:
        for bpe in (1, 2, 4, 99):
            m = fresh()
            w = DataParallel(m, force=True, buckets_per_exchange=bpe, wire_dtype=wire, collective=coll)
            assert w.collective == coll, (w.collective, w._why)
            w._stub = stub
            Rw = [None]
            fw = lambda: bench.step_fn(w, batch, nv, na, Rw)   # noqa: E731
            e_ms, i_ms = median_ms(fw)
            w.begin_step_timing()
            fw()
            comm_ms, nbytes = w.end_step_timing()
            g_txt = "replay: not capturable (send / receive pairs)"
            if coll != "a2a":
                try:
                    gw = GraphedStep(w, fw)
                    g_ms, _ = median_ms(gw, warm=10)
                    del gw
                    g_txt = "replay %.3f ms (+%.3f)" % (g_ms, g_ms - g_plain)
                except Exception as e:  # noqa: BLE001
                    g_txt = "replay: capture failed (%s)" % str(e).replace("\n", " | ")[:200]
            print("DP %-14s wire=%s buckets_per_exchange=%-3s  eager %.3f ms (+%.3f; host issue %.3f)   %s   comm stream busy %.3f ms"
                  % (coll + (" STUB" if stub else ""), "fp32" if wire == torch.float32 else "bf16", "all" if bpe == 99 else bpe, e_ms, e_ms - e_plain, i_ms, g_txt,
                     comm_ms), flush=True)
            m.rt.bucket_hook = None
            m.rt.finish_hook = None
            del w, m
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
