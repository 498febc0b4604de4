#!/usr/bin/env python3
"""bench.py — interval-queries/sec of the TIM encoder hot path (fwd+bwd) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision bf16|fp32] [--batch 64]

Workload (BASELINE.json configs[1] = SURVEY.md C2a): EPIC-100 audio-visual recognition,
d_model 512 (E 1024), 6 layers, 8 heads, 50+50 feature tokens, 15 visual + 10 audio interval
queries per window (S = 155), 64 windows per GPU, training mode with the reference dropout
rates (feat .5 / seq .5 / enc .1), synthetic features, random-init weights of the true shapes.
One step = time_mlp + encoder forward, fixed-cotangent loss (no host loss code), full backward
to every parameter gradient, operand-copy refresh of the weights (as after an optimizer step)
and, for N > 1, the RCCL gradient all-reduce.  Inputs are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tim_amd import synth  # noqa: E402
from tim_amd.config import named_config  # noqa: E402

GFLOP_PER_QUERY = {"C2a": 1.970, "C2b": 2.636, "C3": 1.583, "C1": 0.232}  # BASELINE.md section 4, fwd+bwd
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP32_TFLOPS = 157.3


def build_model(cfg, precision, dev, seed=0):
    from tim_amd.tim import TIM
    m = TIM(cfg.num_class, visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim,
            feat_drop=cfg.feat_drop, seq_drop=cfg.seq_drop, d_model=cfg.d_model,
            feedforward_scale=cfg.feedforward_scale, nhead=cfg.nhead, num_layers=cfg.num_layers,
            enc_dropout=cfg.enc_dropout, input_modality=cfg.input_modality, data_modality=cfg.data_modality,
            num_feats=cfg.num_feats, include_verb_noun=cfg.include_verb_noun, precision=precision)
    sd = synth.make_state_dict(cfg, seed=seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev), sd


def make_batch(cfg, B, nv, na, seed, dev):
    inp = synth.make_inputs(cfg, B, nv, na, seed=seed)
    return {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}


def step_fn(model, batch, nv, na, R):
    inner = model.module if hasattr(model, "module") else model
    for p in inner.parameters():
        p.grad = None
    inner.rt.invalidate_weights()  # weights changed (optimizer step): redo the operand copies
    te = model(batch["times"], "time_mlp")
    (verb, noun, action, audio), feats = model([batch["visual"], batch["audio"]], "encoder", te, nv, na)
    outs = [t for t in (verb, noun, action, audio, feats) if t is not None]
    if R[0] is None:
        g = torch.Generator(device="cpu").manual_seed(1)
        R[:] = [torch.randn(o.shape, generator=g).to(o.device) * 0.05 for o in outs]
    torch.autograd.backward(outs, R)


def gemm_roofline(model, cfg, B, nv, na, precision, reps=20):
    """Per-launch timing of the dominant kernel family (the MFMA NT GEMM) with HIP events on the launch
    stream, for every GEMM shape of one encoder layer; algorithmic FLOPs = 2*M*N*K per launch."""
    from tim_amd import _lib as L
    rt = model.rt
    dev = next(model.parameters()).device
    E, FF = cfg.E, cfg.FF
    S = cfg.F + cfg.num_queries(nv, na)
    M = B * S
    Mp = (M + 63) // 64 * 64
    shapes = [  # name, M, N, K, epi, launches per layer (fwd + bwd)
        ("in_proj fwd", M, 3 * E, E, L.EPI_STORE_T, 1), ("out_proj fwd", M, E, E, L.EPI_DROP_RES_F32, 1),
        ("ffn1 fwd", M, FF, E, L.EPI_GELU_DROP_T2, 1), ("ffn2 fwd", M, E, FF, L.EPI_DROP_RES_F32, 1),
        ("ffn2 dgrad", M, FF, E, L.EPI_DGELU_T, 1), ("ffn1 dgrad", M, E, FF, L.EPI_ADD_F32, 1),
        ("out_proj dgrad", M, E, E, L.EPI_STORE_T, 1), ("in_proj dgrad", M, E, 3 * E, L.EPI_ADD_F32, 1),
        ("in_proj wgrad", 3 * E, E, Mp, "slab", 1), ("out_proj wgrad", E, E, Mp, "slab", 1),
        ("ffn1 wgrad", FF, E, Mp, "slab", 1), ("ffn2 wgrad", E, FF, Mp, "slab", 1),
    ]
    g = torch.Generator().manual_seed(3)
    out = []
    tot_flop = tot_ms = 0.0
    for name, m_, n_, k_, epi, cnt in shapes:
        A = (torch.randn(m_, k_, generator=g)).to(dev).to(rt.op_dtype)
        Bm = (torch.randn(n_, k_, generator=g) * k_ ** -0.5).to(dev).to(rt.op_dtype)
        o0 = torch.zeros((m_, n_), dtype=torch.float32, device=dev)
        o1 = torch.zeros((m_, n_), dtype=torch.float32, device=dev)
        res = torch.zeros((m_, n_), dtype=torch.float32, device=dev)
        bias = torch.zeros(n_, device=dev)
        sk = 1
        if epi == "slab":
            # weight gradient dW[m_, n_] += dY[M, m_]^T X[M, n_] through timhip_wgrad (transposing-read TN
            # MFMA kernel + slab reduce), timed as one unit; M = B*S rows
            dY = torch.randn(M, m_, generator=g).to(dev).to(rt.op_dtype)
            Xa = torch.randn(M, n_, generator=g).to(dev).to(rt.op_dtype)
            dW = torch.zeros((m_, n_), dtype=torch.float32, device=dev)
            dbv = torch.zeros(m_, dtype=torch.float32, device=dev)
            for _ in range(3):
                rt.wgrad(dY, m_, Xa, n_, M, dW, dbv)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                rt.wgrad(dY, m_, Xa, n_, M, dW, dbv)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fl = 2.0 * m_ * n_ * M
            out.append({"gemm": name, "M": m_, "N": n_, "K": M, "us": round(ms * 1e3, 2),
                        "tflops": round(fl / ms / 1e9, 1)})
            tot_flop += fl * cnt
            tot_ms += ms * cnt
            continue
        else:
            kw = dict(out1=o1, ld1=n_, bias=None if epi in (L.EPI_ADD_F32, L.EPI_DGELU_T) else bias,
                      res=res, ldres=n_, aux=o1, ldaux=n_, p_drop=cfg.enc_dropout, seed=7, site=5, splitk=sk)
        for _ in range(3):
            rt.gemm(epi, A, Bm, m_, n_, k_, o0, n_, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            rt.gemm(epi, A, Bm, m_, n_, k_, o0, n_, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * m_ * n_ * k_
        out.append({"gemm": name, "M": m_, "N": n_, "K": k_, "us": round(ms * 1e3, 2),
                    "tflops": round(fl / ms / 1e9, 1)})
        tot_flop += fl * cnt
        tot_ms += ms * cnt
    return out, tot_flop, tot_ms


def cpu_baseline(cfg, sd_np, nv, na, budget_s=12.0):
    """The oracle (torch CPU fp32 restatement, kind "port") timed on this box's host cores on a bounded
    sample of the same workload: fwd+bwd of B=4 windows of the same shapes."""
    from oracle import tim_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))  # torch CPU GEMMs of this size stop scaling (and thrash) beyond ~32 threads
    torch.set_num_threads(cores)
    B = 4
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in sd_np.items()}
    inp = {k: torch.from_numpy(v) for k, v in synth.make_inputs(cfg, B, nv, na, seed=11).items()}

    def one():
        cls, feats = O.forward(sd, cfg, inp["visual"], inp["audio"], inp["times"], nv, na)
        loss = sum(c.sum() for c in cls if c is not None) + feats.sum()
        loss.backward()

    one()
    t0 = time.time()
    n = 0
    while True:
        one()
        n += 1
        if time.time() - t0 > budget_s or n >= 64:
            break
    dt = time.time() - t0
    return {"value": round(B * (nv + na) * n / dt, 2), "unit": "interval-queries/s", "cores": cores, "kind": "port",
            "sample": "oracle/tim_oracle.py fwd+bwd (eval-mode math), %d steps of B=%d windows of the %s shapes, "
                      "torch CPU fp32, %d threads, %.1f s" % (n, B, "C2a", cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--batch", type=int, default=64, help="windows per GPU")
    ap.add_argument("--workload", default="C2a")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py: --gpus %d must be launched with torch.distributed.run (WORLD_SIZE unset)" % args.gpus,
              file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = named_config(args.workload)
    nv, na = (10, 0) if args.workload == "C1" else (15, 10)
    B = args.batch
    model, sd_np = build_model(cfg, args.precision, dev, seed=0)
    model.train()
    run_model = model
    if world > 1:
        from tim_amd.dp import DataParallel
        run_model = DataParallel(model)
    batch = make_batch(cfg, B, nv, na, seed=100 + rank, dev=dev)  # each rank its own shard of windows
    R = [None]

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step_fn(run_model, batch, nv, na, R)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_fn(run_model, batch, nv, na, R)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    queries_per_step = world * B * (nv + na)
    value = queries_per_step * args.steps / dt
    gq = GFLOP_PER_QUERY.get(args.workload)
    out = {
        "metric": "interval-queries/sec (fwd+bwd), d=512 L=6 EPIC-100 window",
        "value": round(value, 1), "unit": "interval-queries/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32",
        "data": "synthetic",
        "config": {"workload": "%s: EPIC-100 A+V recognition, d_model 512 (E 1024), 6 layers, 8 heads, 50+50 "
                               "feature tokens, 15+10 interval queries, train-mode dropout" % args.workload
                   if args.workload == "C2a" else args.workload,
                   "windows_per_gpu": B, "tokens_per_window": cfg.F + cfg.num_queries(nv, na),
                   "global_batch": world * B, "parallelism": "dp%d" % world, "precision": args.precision},
    }
    if gq is not None:
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_FP32_TFLOPS
        out["whole_step"] = {"tflops_algorithmic": round(value / world * gq / 1e3, 1),
                             "frac_of_mfma_peak": round(value / world * gq / 1e3 / peak, 4)}
    if rank == 0 and not args.no_roofline:
        per, fl, ms = gemm_roofline(model, cfg, B, nv, na, args.precision)
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_FP32_TFLOPS
        ach = fl / ms / 1e9
        traffic = None  # HBM-side bytes per launch of the GEMM kernels, from the committed rocprofv3 PMC passes
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
            if args.precision == "bf16" and args.workload == "C2a" and B == 64:
                nt, tn = tj["gemm_nt_bf16_kernel"], tj["wgrad_tn_bf16_kernel"]
                traffic = round((8 * nt["bytes_per_launch"] + 4 * tn["bytes_per_launch"]) / 12.0)
        except Exception:
            traffic = None
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(ach / peak, 4), "traffic": traffic,
                           "traffic_note": "average HBM-side bytes per GEMM launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, "
                                           "profiles/r01_pmc_traffic.json, tools/pmc_traffic.py); algorithmic operand+result "
                                           "bytes average 70 MB per launch",
                           "kernel": "MFMA GEMM family: gemm_nt_%s_kernel (8 launches) + wgrad_tn kernel incl. its slab "
                                     "reduce (4 launches) = the 12 GEMMs of one encoder layer fwd+bwd, 2*M*N*K "
                                     "algorithmic FLOPs each, HIP-event timed on the launch stream" % (
                                         "bf16" if args.precision == "bf16" else "f32"),
                           "per_shape": per}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, sd_np, nv, na)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
