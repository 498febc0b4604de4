#!/usr/bin/env python3
"""Where the logit error of a 16-bit operand mode comes from (CPU, oracle only - a measurement script, not product code).

Evaluates oracle/tim_oracle.py in fp64 with the operands of selected GEMM sites rounded to fp16 / bf16 (what an MFMA kernel
with 16-bit operands and fp32 accumulation does) and prints max / rms |dlogit| against the exact evaluation.
    python tools/err_budget.py [--dist synth|trained] [--dtype fp16|bf16] [--batch 2]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tim_oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402
from tim_amd.config import named_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dist", default="synth")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--config", default="C2a")
    ap.add_argument("--rows", type=float, nargs=2, default=(2.0, 4.0), help="trained: row scale range of in_proj / linear1")
    ap.add_argument("--gain", type=float, nargs=2, default=(0.5, 3.0), help="trained: LayerNorm gain range")
    ap.add_argument("--sigma", type=float, default=0.5, help="trained: log-normal sigma of the feature magnitudes")
    ap.add_argument("--short", action="store_true")
    a = ap.parse_args()
    rd = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    cfg = named_config(a.config)
    nv, na = 15, 10
    sd, inp = H.synth_torch(cfg, a.batch, nv, na, seed=2, dtype=torch.float32)
    if a.dist == "trained":
        from tests.test_gpu_train_parity import trained_like
        sd, inp = trained_like(cfg, sd, inp, row_scale=tuple(a.rows), gain=tuple(a.gain), sigma=a.sigma)
    sd = {k: v.double() for k, v in sd.items()}
    inp = {k: v.double() for k, v in inp.items()}
    wid = {id(v): k for k, v in sd.items()}
    lin0 = O._lin
    state = {"sites": None, "what": "both"}   # sites: callable(name) -> bool; what: both | w | x

    def lin(x, w, b, rd_=None):
        name = wid.get(id(w), "?")
        if state["sites"] is not None and state["sites"](name):
            if state["what"] in ("both", "x"):
                x = x.to(rd).to(torch.float64)
            if state["what"] in ("both", "w"):
                w = w.to(rd).to(torch.float64)
        return lin0(x, w, b, None)

    O._lin = lin

    def run(sites, what="both", attn_rd=False):
        state["sites"], state["what"] = sites, what
        with torch.no_grad():
            if attn_rd:   # attention internals rounded as the MFMA attention kernel does (q, k, v, p): use the oracle's own rd path
                O._lin = lin0
                cls, _ = O.forward(sd, cfg, inp["visual"], inp["audio"], inp["times"], nv, na, rd=rd)
                O._lin = lin
            else:
                cls, _ = O.forward(sd, cfg, inp["visual"], inp["audio"], inp["times"], nv, na)
        return torch.cat([c.flatten() for c in cls if c is not None])

    ref = run(None)
    print("config %s, dist %s, %s operands; |logit| max %.2f rms %.2f" % (a.config, a.dist, a.dtype, ref.abs().max(), ref.square().mean().sqrt()))

    def rep(label, sites, what="both", attn_rd=False):
        d = run(sites, what, attn_rd) - ref
        print("%-58s max %.3e  rms %.3e" % (label, d.abs().max(), d.square().mean().sqrt()))

    layer = lambda n: ".layers." in n
    rep("all GEMM operands + attention internals", None, attn_rd=True)
    rep("all GEMM sites (no attention-internal rounding)", lambda n: True)
    rep("layers + embedders (the fp16 mode: time MLP and heads split)", lambda n: layer(n) or "embedder" in n)
    if a.short:
        return
    rep("time MLP", lambda n: n.startswith("time_mlp"))
    rep("embedders", lambda n: "embedder" in n)
    rep("heads", lambda n: n.startswith("cls_head"))
    rep("encoder layers, all four Linears", layer)
    rep("  in_proj (both operands)", lambda n: n.endswith("in_proj_weight"))
    rep("  in_proj weights only", lambda n: n.endswith("in_proj_weight"), "w")
    rep("  in_proj activations only", lambda n: n.endswith("in_proj_weight"), "x")
    rep("  out_proj", lambda n: n.endswith("out_proj.weight"))
    rep("  linear1", lambda n: n.endswith("linear1.weight"))
    rep("  linear2", lambda n: n.endswith("linear2.weight"))
    rep("layers except in_proj + embedders", lambda n: (layer(n) and not n.endswith("in_proj_weight")) or "embedder" in n)
    # the fp16 mode with out_proj's WEIGHTS exact (split [hi | lo], round 3): activations of out_proj still rounded
    state_w = lambda n: (layer(n) and not n.endswith("out_proj.weight")) or "embedder" in n
    rep("fp16 mode, out_proj weights exact (approx: others both + out_proj x)", lambda n: True if state_w(n) else False)
    for l in range(cfg.num_layers):
        rep("  layer %d only" % l, lambda n, l=l: (".layers.%d." % l) in n)


if __name__ == "__main__":
    main()
