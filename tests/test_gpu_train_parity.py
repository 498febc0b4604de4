"""Train-mode parity: the arithmetic bench.py times (model.train(): feature / sequence / attention / residual / FFN dropout)
against the CPU oracle evaluated with THE SAME keep-masks.

The reference draws its masks from torch's global Philox stream, which no other implementation reproduces; here a mask is a
pure function of (step seed, site, element index).  `timhip_dropout_mask` states that function on its own (include/timhip.h),
so the tests (1) pin every site's kernel to it bit for bit - kept set, 1/(1-p) scale, keep rate - and (2) hand the masks of a
whole training forward to `oracle.tim_oracle.forward(..., masks=...)` (rec encodings.py:140-153,249; transformers.py:104-109;
F.multi_head_attention_forward's dropout on the softmax output) and compare logits and every gradient as a model.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tim_oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.test_gpu_parity import build as _build, maxerr, relerr, amax  # noqa: E402
from tim_amd import _lib as L  # noqa: E402
from tim_amd.config import named_config  # noqa: E402
from tim_amd.functional import Runtime, _ru  # noqa: E402

DEV = "cuda:0"


def build(cfg, precision, sd):
    """the model's dropout key derives from torch.initial_seed() when the model is built (functional.Runtime): pinned here to
    torch's default seed value, so that the masks - and the error realisation the asserts see - do not depend on which tests
    ran before (other tests call torch.manual_seed).  The logit error of the fp16 mode in training mode varies with the draw:
    8.1e-4 on this key, 9.5e-4 on the key the full suite's order used to produce (C2a, B = 64)."""
    torch.manual_seed(67280421310721)
    return _build(cfg, precision, sd)


def st():
    return torch.cuda.current_stream().cuda_stream


def keep_mask(seed, site, p, rows, cols):
    """[rows, cols] uint8 on the CPU: the library's statement of the site's mask (element (r, c) = linear index r * cols + c)"""
    mk = torch.empty((rows, cols), dtype=torch.uint8, device=DEV)
    L.call("timhip_dropout_mask", seed, site, float(p), rows, cols, L.ptr(mk), st())
    torch.cuda.synchronize()
    return mk.cpu()


def rate_ok(mask, p, sigmas=4.0):
    n = mask.numel()
    keep = mask.double().mean().item()
    return abs(keep - (1.0 - p)) <= sigmas * math.sqrt(p * (1 - p) / n) + 1.0 / 65536   # (p is quantised to 1/65536)


def site_masks(cfg, seed, B, S, inp):
    """the keep-masks of one training forward, keyed as oracle.tim_oracle expects them"""
    E, FF, Hh, F = cfg.E, cfg.FF, cfg.nhead, cfg.F
    M = B * S
    masks = {}
    for name, site in (("visual", L.SITE_FEAT_V), ("audio", L.SITE_FEAT_A)):
        x = inp[name]
        if x.dim() != 3:
            continue
        Cin = x.shape[2]
        cq = (Cin + 3) // 4 * 4   # the feature dropout numbers its elements over rows of ceil(C / 4) quads
        masks["feat_" + name] = keep_mask(seed, site, cfg.feat_drop, B * cfg.num_feats, cq)[:, :Cin].reshape(B, cfg.num_feats, Cin)
    masks["seq"] = keep_mask(seed, L.SITE_SEQ, cfg.seq_drop, M, E).reshape(B, S, E)
    LP = (F + 1 + 7) // 8 * 8
    for l in range(cfg.num_layers):
        masks["l%d_attn" % l] = keep_mask(seed, L.layer_site(l, L.SITE_L_ATTN), cfg.enc_dropout, B * Hh * S, LP) \
            .reshape(B, Hh, S, LP)[..., :F + 1]
        masks["l%d_drop1" % l] = keep_mask(seed, L.layer_site(l, L.SITE_L_DROP1), cfg.enc_dropout, M, E).reshape(B, S, E)
        masks["l%d_ffn" % l] = keep_mask(seed, L.layer_site(l, L.SITE_L_FFN), cfg.enc_dropout, M, FF).reshape(B, S, FF)
        masks["l%d_drop2" % l] = keep_mask(seed, L.layer_site(l, L.SITE_L_DROP2), cfg.enc_dropout, M, E).reshape(B, S, E)
    return masks


# ------------------------------------------------------------------------------------------------
# (a) every dropout site's kernel == timhip_dropout_mask, bit for bit
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("R,Cin,p", [(100, 1024, 0.5), (37, 2304, 0.5), (12, 24, 0.3), (9, 41, 0.1)])
def test_feature_dropout_site(prec, R, Cin, p):
    """timhip_cast_rows (encodings.py:21 nn.Dropout(feat_drop) in front of the embedder Linear) and its backward
    timhip_dropout_rows_bwd: kept set, scale 1 / (1 - p), keep rate"""
    rt = Runtime(prec)
    seed, site = 0x1234567, L.SITE_FEAT_A
    x = (torch.rand(R, Cin) + 0.5).to(DEV)
    ld = _ru(Cin)
    xT = torch.full((R, ld), 7.0, dtype=rt.op_dtype, device=DEV)
    L.call("timhip_cast_rows", rt.prec, L.ptr(x), R, Cin, Cin, L.ptr(xT), ld, p, seed, site, None, st())
    cq = (Cin + 3) // 4 * 4
    mk = keep_mask(seed, site, p, R, cq)[:, :Cin]
    got = xT.float().cpu()
    assert (got[:, Cin:] == 0).all()
    assert torch.equal(got[:, :Cin] != 0, mk.bool())
    want = x.cpu() * mk.float() / (1 - p)
    assert (got[:, :Cin] - want).abs().max().item() <= {"fp32": 1e-6, "fp16": 2e-3, "bf16": 2e-2}[prec]
    assert rate_ok(mk, p)
    g = (torch.rand(R, Cin) + 0.5).to(DEV)
    dx = torch.empty((R, Cin), device=DEV)
    L.call("timhip_dropout_rows_bwd", L.ptr(g), R, Cin, Cin, L.ptr(dx), Cin, p, seed, site, st())
    torch.cuda.synchronize()
    assert (dx.cpu() - g.cpu() * mk.float() / (1 - p)).abs().max().item() <= 1e-6


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
@pytest.mark.parametrize("B,nf,nq,d,p", [(3, 6, 5, 32, 0.25), (2, 50, 25, 512, 0.5)])
def test_sequence_dropout_site(prec, B, nf, nq, d, p):
    """timhip_assemble_fwd (encodings.py:249 self.dropout on the concatenated sequence): kept set, scale, keep rate; the
    backward regenerates the same mask (d_e = mask / (1 - p) * dx on the embedded-feature half)"""
    rt = Runtime(prec)
    S, T, E = nf + nq, nf + nq, 2 * d
    rows = [(0, s, s, -1) for s in range(nf)] + [(1, 0, nf + j, -1) for j in range(nq)]
    tab = torch.tensor(rows, dtype=torch.int32).to(DEV)
    e0 = (torch.rand(B * nf, d) + 0.5).to(DEV)
    cls = (torch.rand(1, d) + 0.5).to(DEV)
    te = (torch.rand(B, T, d) + 0.5).to(DEV)
    x = torch.empty((B * S, E), device=DEV)
    xt = torch.empty((B * S, E), dtype=rt.op_dtype, device=DEV)
    seed = 987654321
    L.call("timhip_assemble_fwd", rt.prec, L.ptr(tab), B, S, d, L.ptr(e0), None, nf, L.ptr(cls), L.ptr(te), T, None, p, seed,
           L.SITE_SEQ, L.ptr(x), L.ptr(xt), st())
    mk = keep_mask(seed, L.SITE_SEQ, p, B * S, E)
    full = torch.cat([torch.cat([e0.view(B, nf, d), cls.expand(B, nq, d)], 1), te], -1).reshape(B * S, E).cpu()
    want = full * mk.float() / (1 - p)
    assert torch.equal(x.cpu() != 0, mk.bool())
    assert (x.cpu() - want).abs().max().item() <= 1e-6
    assert (xt.float().cpu() - want).abs().max().item() <= {"fp32": 1e-6, "fp16": 2e-3}[prec]
    assert rate_ok(mk, p)
    dx = (torch.rand(B * S, E) + 0.5).to(DEV)
    d_e0 = torch.empty((B * nf, d), device=DEV)
    d_cls = torch.zeros((1, d), device=DEV)
    d_te = torch.empty((B, T, d), device=DEV)
    d_mod = torch.zeros((1, E), device=DEV)
    L.call("timhip_assemble_bwd", L.ptr(tab), B, S, d, L.ptr(dx), nf, T, p, seed, L.SITE_SEQ, L.ptr(d_e0), None, L.ptr(d_cls),
           L.ptr(d_te), L.ptr(d_mod), st())
    torch.cuda.synchronize()
    gm = (dx.cpu() * mk.float() / (1 - p)).view(B, S, E)
    assert (d_e0.cpu().view(B, nf, d) - gm[:, :nf, :d]).abs().max().item() <= 1e-6
    assert (d_te.cpu() - gm[:, :, d:]).abs().max().item() <= 1e-5
    assert (d_cls.cpu()[0] - gm[:, nf:, :d].sum((0, 1))).abs().max().item() <= 1e-4 * B * nq


@pytest.mark.parametrize("prec", ["fp16", "bf16", "fp32"])
def test_layer_forward_keep_bits_and_sites(prec):
    """timhip_layer_fwd in training mode (transformers.py:104-109): the FFN keep-bits LayerNorm-1 writes into the saved block
    equal timhip_dropout_mask(seed, site FFN of that layer); the saved hidden activations h are zero exactly where the mask
    drops (h = mask / (1 - p) * gelu(linear1): the bits were APPLIED, not only stored)"""
    cfg = H.tiny_cfg("recognition", "audio_visual", "audio_visual", True)
    cfg.d_model, cfg.nhead = 64, 2          # FF = 256, E = 128
    cfg.enc_dropout, cfg.feat_drop, cfg.seq_drop = 0.2, 0.0, 0.0
    nv, na, B = 4, 2, 3
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=7, dtype=torch.float32)
    m = build(cfg, prec, sd).train()
    te = m(inp["times"].to(DEV), "time_mlp")
    cls, feats = m([inp["visual"].to(DEV), inp["audio"].to(DEV)], "encoder", te, nv, na)
    node = cls[2].grad_fn
    seed = m.rt.last_seed
    S = node.dims[3]
    M, FF = B * S, cfg.FF
    desc = L.TimDesc(B, S, cfg.F, cfg.d_model, cfg.E, cfg.nhead, FF, m.rt.prec, cfg.enc_dropout, seed, 0, 0, None)
    off, nb = C.c_size_t(), C.c_size_t()
    for l in range(cfg.num_layers):
        sv = node.layer_saved[l]
        L.call("timhip_layer_saved_field", C.byref(desc), L.SAVED_FFN_KEEP_BITS, C.byref(off), C.byref(nb))
        bits = sv[off.value:off.value + nb.value].cpu().numpy().reshape(M, FF // 8)
        got = torch.from_numpy(np.unpackbits(bits, axis=1, bitorder="little"))
        mk = keep_mask(seed, L.layer_site(l, L.SITE_L_FFN), cfg.enc_dropout, M, FF)
        assert torch.equal(got, mk), l
        assert rate_ok(mk, cfg.enc_dropout)
        L.call("timhip_layer_saved_field", C.byref(desc), L.SAVED_H, C.byref(off), C.byref(nb))
        h = sv[off.value:off.value + nb.value].view(m.rt.op_dtype).view(M, FF).float().cpu()
        assert (h[mk == 0] == 0).all(), l
        assert (h[mk == 1] != 0).float().mean().item() > 0.99, l   # gelu(u) is exactly 0 only at u = 0 (or fp16 underflow)


def test_attention_keep_bits_are_the_mask(monkeypatch):
    """round 6: timhip_attn_keep_bits draws the attention dropout of every layer ahead of the stack, into the layers' saved blocks
    (two 64-bit words per (window, head, token row); word g, bit 4 c + t = key 8 c + 4 g + t).  (a) the words are
    timhip_dropout_mask's statement of the site, bit for bit, for two layers in one launch; (b) the C2a model's training step
    with the bits (the default) and with the kernels drawing their own masks (TIM_AMD_ATTN_KEEP_BITS=0) gives IDENTICAL logits
    and parameter gradients - the forward and the fused backward read the same decisions they used to compute."""
    cfg = named_config("C2a")
    B, S, F, Hh, E = 3, 155, cfg.F, cfg.nhead, cfg.E
    seed, p = 0x1D2C3B4A5968, cfg.enc_dropout
    desc = L.TimDesc(B, S, F, cfg.d_model, E, Hh, cfg.FF, L.PREC_F16, p, seed, 0, 0, None)
    nb = L.load().timhip_layer_saved_bytes(C.byref(desc))
    saved = [torch.zeros(nb, dtype=torch.uint8, device=DEV) for _ in range(2)]
    L.call("timhip_attn_keep_bits", C.byref(desc), 2, (C.c_void_p * 2)(*[t.data_ptr() for t in saved]), st())
    off, nbytes = C.c_size_t(), C.c_size_t()
    L.call("timhip_layer_saved_field", C.byref(desc), L.SAVED_ATTN_KEEP_BITS, C.byref(off), C.byref(nbytes))
    rows, LP = B * Hh * S, (F + 1 + 7) // 8 * 8
    assert nbytes.value == rows * 16
    for l in range(2):
        words = saved[l][off.value:off.value + nbytes.value].cpu().numpy().view(np.uint64).reshape(rows, 2)
        mk = keep_mask(seed, L.layer_site(l, L.SITE_L_ATTN), p, rows, LP).numpy()
        for k in range(F):   # (keys F .. : the self slot and the padding - drawn by the kernels themselves / never used)
            c, e = k >> 3, k & 7
            got = (words[:, e >> 2] >> np.uint64(4 * c + (e & 3))) & np.uint64(1)
            assert np.array_equal(got.astype(np.uint8), mk[:, k]), (l, k)
    assert not np.array_equal(saved[0][off.value:off.value + 64].cpu().numpy(), saved[1][off.value:off.value + 64].cpu().numpy())
    desc32 = L.TimDesc(B, S, F, cfg.d_model, E, Hh, cfg.FF, L.PREC_FP32, p, seed, 0, 0, None)
    assert L.load().timhip_attn_keep_bits(C.byref(desc32), 2, (C.c_void_p * 2)(*[t.data_ptr() for t in saved]), st()) == L.EUNSUPPORTED
    # (b) the model, both ways
    nv, na, Bm = 15, 10, 2
    sd, inp = H.synth_torch(cfg, Bm, nv, na, seed=2, dtype=torch.float32)
    with torch.no_grad():
        o_eval = H.named_outputs(*O.forward(sd, cfg, inp["visual"], inp["audio"], inp["times"], nv, na))
    R = H.cotangents(cfg, Bm, nv, na, o_eval, seed=2, dtype=torch.float32)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("TIM_AMD_ATTN_KEEP_BITS", flag)
        res[flag] = run_train(build(cfg, "fp16", sd), inp, nv, na, R)
    assert res["1"]["seed"] == res["0"]["seed"]
    for k in res["1"]["outs"]:
        assert torch.equal(res["1"]["outs"][k], res["0"]["outs"][k]), k
    for k in res["1"]["grads"]:   # (a few gradients end in atomics - cls tokens, embedder LayerNorms: equal up to their summation order)
        a_, b_ = res["1"]["grads"][k], res["0"]["grads"][k]
        assert (a_ - b_).abs().max().item() <= 2e-6 * max(b_.abs().max().item(), 1e-30), k
    assert torch.equal(res["1"]["grads"]["transformer_encoder.layers.0.self_attn.in_proj_weight"],
                       res["0"]["grads"]["transformer_encoder.layers.0.self_attn.in_proj_weight"])


# ------------------------------------------------------------------------------------------------
# (b) whole model in .train() against the oracle fed with the same masks
# ------------------------------------------------------------------------------------------------
def run_train(m, inp, nv, na, R, step=100):
    m.train()
    m.rt.step = step
    vis = inp["visual"].to(DEV).float()
    aud = inp["audio"].to(DEV).float()
    times = inp["times"].to(DEV).float().requires_grad_(True)
    if vis.dim() == 3:
        vis.requires_grad_(True)
    if aud.dim() == 3:
        aud.requires_grad_(True)
    te = m(times, "time_mlp")
    cls, feats = m([vis, aud], "encoder", te, nv, na)
    outs = H.named_outputs(cls, feats)
    seed = m.rt.last_seed
    S = cls[2].grad_fn.dims[3] if cls[2] is not None else cls[3].grad_fn.dims[3]
    loss = sum((outs[k] * R[k].to(DEV)).sum() for k in outs)
    loss.backward()
    torch.cuda.synchronize()
    return dict(outs={k: v.detach().cpu() for k, v in outs.items()},
                grads={k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None},
                gin={k: t.grad.detach().cpu() for k, t in (("visual", vis), ("audio", aud), ("times", times)) if t.grad is not None},
                seed=seed, S=S)


def oracle_train(cfg, sd, inp, nv, na, R, masks, dtype=torch.float32):
    sd = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    leaves = {k: inp[k].to(dtype).clone().requires_grad_(inp[k].dim() == 3) for k in ("visual", "audio", "times")}
    cls, feats = O.forward(sd, cfg, leaves["visual"], leaves["audio"], leaves["times"], nv, na, masks=masks)[:2]
    outs = H.named_outputs(cls, feats)
    sum((outs[k] * R[k].to(dtype)).sum() for k in outs).backward()
    return ({k: v.detach() for k, v in outs.items()}, {k: v.grad for k, v in sd.items() if v.grad is not None},
            {k: v.grad for k, v in leaves.items() if v.grad is not None})


def grad_agreement(got, want):
    a, b = got.double().flatten(), want.double().flatten()
    cos = (a @ b / (a.norm() * b.norm() + 1e-300)).item() if b.norm() > 0 else 1.0
    return cos, relerr(got, want)


@pytest.mark.parametrize("fname,im,dm,vn,nv,na", H.rec_golden_cases())
def test_tiny_train_mode_fp32_vs_oracle_with_masks(fname, im, dm, vn, nv, na):
    """every modality combination, fp32 kernels, all five kinds of dropout on: logits / feats <= 1e-5, every parameter and
    input gradient <= 1e-4 relative"""
    cfg = H.tiny_cfg("recognition", im, dm, vn)
    cfg.feat_drop, cfg.seq_drop, cfg.enc_dropout = 0.3, 0.25, 0.2
    B = 3
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=5, dtype=torch.float32)
    with torch.no_grad():
        o_eval = H.named_outputs(*O.forward(sd, cfg, inp["visual"], inp["audio"], inp["times"], nv, na))
    R = H.cotangents(cfg, B, nv, na, o_eval, seed=5, dtype=torch.float32)
    m = build(cfg, "fp32", sd)
    res = run_train(m, inp, nv, na, R)
    masks = site_masks(cfg, res["seed"], B, res["S"], inp)
    o, g, gin = oracle_train(cfg, sd, inp, nv, na, R, masks, torch.float64)
    assert any(maxerr(o[k], o_eval[k]) > 1e-2 for k in o)   # dropout really changed the outputs
    for k, v in res["outs"].items():
        assert maxerr(v, o[k]) <= 1e-5 * max(1.0, amax(o[k])), (k, maxerr(v, o[k]))
    for k, v in g.items():
        if k.startswith("drloc_mlp"):
            continue
        assert relerr(res["grads"][k], v) <= 1e-4, (k, relerr(res["grads"][k], v))
    for k, v in gin.items():
        assert relerr(res["gin"][k], v) <= 1e-4, (k, relerr(res["gin"][k], v))


@pytest.mark.parametrize("prec,B", [("fp32", 2), ("fp16", 2)])
def test_c2a_train_mode_vs_oracle_with_masks(prec, B):
    """C2a (the headline model, reference dropout rates 0.5 / 0.5 / 0.1) in .train(), B = 2: fp32 kernels <= 1e-5 on the
    logits, fp16 <= 1e-3; every parameter gradient cos >= 0.9995 and within 6e-2 of the tensor's largest element (fp16),
    1e-4 (fp32)"""
    cfg = named_config("C2a")
    nv, na = 15, 10
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=2, dtype=torch.float32)
    with torch.no_grad():
        o_eval = H.named_outputs(*O.forward(sd, cfg, inp["visual"], inp["audio"], inp["times"], nv, na))
    R = H.cotangents(cfg, B, nv, na, o_eval, seed=2, dtype=torch.float32)
    m = build(cfg, prec, sd)
    res = run_train(m, inp, nv, na, R)
    masks = site_masks(cfg, res["seed"], B, res["S"], inp)
    o, g, gin = oracle_train(cfg, sd, inp, nv, na, R, masks)
    tol = {"fp32": 1e-5, "fp16": 1e-3}[prec]
    worst = 0.0
    for k, v in res["outs"].items():
        if k == "feats":
            continue
        e = maxerr(v, o[k]) / (max(1.0, amax(o[k])) if prec == "fp32" else 1.0)
        worst = max(worst, e)
        assert e <= tol, (prec, k, e)
    wc, wr = 1.0, 0.0
    for k, v in g.items():
        if k.startswith("drloc_mlp"):
            continue
        cos, rel = grad_agreement(res["grads"][k], v)
        wc, wr = min(wc, cos), max(wr, rel)
        assert cos >= 0.9995, (prec, k, cos)
        assert rel <= (1e-3 if prec == "fp32" else 6e-2), (prec, k, rel)
    print("C2a train mode B=%d %s: worst |dlogit| %.3g, min grad cos %.6f, max grad rel err %.3g" % (B, prec, worst, wc, wr))


_T64 = {}


def _c2a_b64_train_oracle(seed):
    """fp32 CPU oracle of the training step bench.py times (64 windows, dropout 0.5 / 0.5 / 0.1) under the masks of `seed`"""
    if seed not in _T64:
        cfg = named_config("C2a")
        B, nv, na = 64, 15, 10
        sd, inp = H.synth_torch(cfg, B, nv, na, seed=2, dtype=torch.float32)
        with torch.no_grad():
            o_eval = H.named_outputs(*O.forward(sd, cfg, inp["visual"], inp["audio"], inp["times"], nv, na))
        R = H.cotangents(cfg, B, nv, na, o_eval, seed=2, dtype=torch.float32)
        S = cfg.F + cfg.num_queries(nv, na)
        masks = site_masks(cfg, seed, B, S, inp)
        o, g, gin = oracle_train(cfg, sd, inp, nv, na, R, masks)
        _T64.clear()
        _T64[seed] = dict(cfg=cfg, sd=sd, inp=inp, R=R, o=o, g=g, gin=gin)
    return _T64[seed]


def test_c2a_production_batch_train_mode_fp16():
    """THE STEP THE BENCH TIMES: C2a, B = 64 windows (M = 9920 rows: ping-pong GEMM tiles, grouped weight gradients, chained
    layers), fp16 mode, .train() with the reference's dropout rates - against the fp32 CPU oracle under the same masks.
    Logits <= 1e-3 over all 4.06 M of them; every parameter gradient ELEMENTWISE: cos >= 0.9995, max error <= 6e-2 of the
    tensor's largest element."""
    cfg = named_config("C2a")
    B, nv, na = 64, 15, 10
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=2, dtype=torch.float32)
    m = build(cfg, "fp16", sd)
    import tim_amd.synth as synth
    nc = cfg.num_class
    shapes = {"verb": (B * nv, nc[0][0]), "noun": (B * nv, nc[0][1]), "action": (B * nv, nc[0][2]), "audio": (B * na, nc[1]),
              "feats": (B, cfg.F, cfg.E)}
    R = {k: torch.from_numpy(v).float() for k, v in synth.make_cotangents(cfg, B, nv, na, shapes, seed=2, dtype=np.float64).items()}
    # run the real step first, then ask the runtime which Philox key it used: the oracle gets that step's masks
    res = run_train(m, inp, nv, na, R)
    c = _c2a_b64_train_oracle(res["seed"])
    for k in R:
        assert torch.equal(R[k], c["R"][k])   # same cotangents on both sides
    worst = 0.0
    n = 0
    for k, v in res["outs"].items():
        if k == "feats":
            continue
        e = maxerr(v, c["o"][k])
        n += v.numel()
        worst = max(worst, e)
        assert e <= 1e-3, (k, e)
    wc, wr = 1.0, 0.0
    for k, v in c["g"].items():
        if k.startswith("drloc_mlp"):
            continue
        cos, rel = grad_agreement(res["grads"][k], v)
        wc, wr = min(wc, cos), max(wr, rel)
        assert torch.isfinite(res["grads"][k]).all(), k
        assert cos >= 0.9995, (k, cos)
        assert rel <= 6e-2, (k, rel)
    for k, v in c["gin"].items():
        assert relerr(res["gin"][k], v) <= 6e-2, (k, relerr(res["gin"][k], v))
    print("C2a B=64 fp16 TRAIN mode: worst |dlogit| %.3g over %d logits; gradients min cos %.6f, max rel err %.3g" % (worst, n, wc, wr))


@pytest.mark.parametrize("B", [8, 37])
def test_c2a_train_mode_fp16_operating_points_of_the_drop_in(B):
    """The full C2a model in .train(), fp16 mode, at the two batch sizes the reference's own loader produces besides 64:
    B = 8 - the published recipe (global batch 64, utils/parser.py:87) on 8 GPUs, datasets/loader.py:48 divides the batch by the
    GPU count (M = 1240 rows: below the one-block-per-CU GEMM kernels' 192-tile threshold, i.e. the small-problem kernels,
    the split weight-gradient path, a one-wave-per-row-block attention grid of 64 blocks) - and a ragged tail batch B = 37
    (loader.py:58 drop_last=False; M = 5735 rows: neither the 160- nor the 128-row tile divides it, 1850 embedder rows are not a
    multiple of the paired LayerNorm's 16-row blocks).  Same bar as B = 64: every logit within 1e-3 of the fp32 CPU oracle under
    the step's masks; every parameter gradient elementwise cos >= 0.9995 and within 6e-2 of the tensor's largest element."""
    cfg = named_config("C2a")
    nv, na = 15, 10
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=2, dtype=torch.float32)
    m = build(cfg, "fp16", sd)
    import tim_amd.synth as synth
    nc = cfg.num_class
    shapes = {"verb": (B * nv, nc[0][0]), "noun": (B * nv, nc[0][1]), "action": (B * nv, nc[0][2]), "audio": (B * na, nc[1]),
              "feats": (B, cfg.F, cfg.E)}
    R = {k: torch.from_numpy(v).float() for k, v in synth.make_cotangents(cfg, B, nv, na, shapes, seed=2, dtype=np.float64).items()}
    res = run_train(m, inp, nv, na, R)
    masks = site_masks(cfg, res["seed"], B, res["S"], inp)
    o, g, gin = oracle_train(cfg, sd, inp, nv, na, R, masks)
    worst, n = 0.0, 0
    for k, v in res["outs"].items():
        if k == "feats":
            continue
        e = maxerr(v, o[k])
        n += v.numel()
        worst = max(worst, e)
        assert e <= 1e-3, (B, k, e)
    wc, wr = 1.0, 0.0
    for k, v in g.items():
        if k.startswith("drloc_mlp"):
            continue
        cos, rel = grad_agreement(res["grads"][k], v)
        wc, wr = min(wc, cos), max(wr, rel)
        assert torch.isfinite(res["grads"][k]).all(), k
        assert cos >= 0.9995, (B, k, cos)
        assert rel <= 6e-2, (B, k, rel)
    for k, v in gin.items():
        assert relerr(res["gin"][k], v) <= 6e-2, (B, k, relerr(res["gin"][k], v))
    print("C2a B=%d fp16 TRAIN mode: worst |dlogit| %.3g over %d logits; gradients min cos %.6f, max rel err %.3g" % (B, worst, n, wc, wr))


_DRAWS = [(1, 7), (20260930, 1234), (0x5EED5EED5EED, 99991), (2, 1), (3, 100), (17, 4242), (99, 31337), (123456789, 5),
          (0xC0FFEE, 77), (0xBADC0DE, 2025), (31415926, 8), (27182818, 65536)]
_DRAW_ERRS = {}


@pytest.mark.parametrize("key,step", _DRAWS)
def test_c2a_production_batch_train_mode_fp16_other_draws(key, step):
    """the 1e-3 claim as a statement about the DISTRIBUTION of draws, not one pinned key: the same B = 64 training forward
    under twelve unrelated (model key, step) pairs - different masks at every site - each against the fp32 oracle under ITS
    masks.  Forward only (the gradients' agreement does not depend on the draw: see the pinned case above)."""
    cfg = named_config("C2a")
    B, nv, na = 64, 15, 10
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=2, dtype=torch.float32)
    torch.manual_seed(key)
    m = _build(cfg, "fp16", sd).train()
    m.rt.step = step
    with torch.no_grad():
        te = m(inp["times"].to(DEV).float(), "time_mlp")
        cls, feats = m([inp["visual"].to(DEV).float(), inp["audio"].to(DEV).float()], "encoder", te, nv, na)
    seed = m.rt.last_seed
    outs = {k: v.cpu() for k, v in H.named_outputs(cls, feats).items()}
    S = cfg.F + cfg.num_queries(nv, na)
    masks = site_masks(cfg, seed, B, S, inp)
    with torch.no_grad():
        o = H.named_outputs(*O.forward(sd, cfg, inp["visual"], inp["audio"], inp["times"], nv, na, masks=masks))
    errs = {k: maxerr(outs[k], o[k]) for k in outs if k != "feats"}
    print("C2a B=64 fp16 TRAIN mode, key %d step %d (seed %d): worst |dlogit| %.3g  %s"
          % (key, step, seed, max(errs.values()), {k: "%.2e" % v for k, v in errs.items()}))
    _DRAW_ERRS[(key, step)] = max(errs.values())
    if len(_DRAW_ERRS) == len(_DRAWS):   # the tail of the distribution in one line (round-4 review: four draws say little)
        v = sorted(_DRAW_ERRS.values())
        print("C2a B=64 fp16 TRAIN mode, %d draws: max |dlogit| min %.3g median %.3g MAX %.3g (bound 1e-3)"
              % (len(v), v[0], v[len(v) // 2], v[-1]))
    assert max(errs.values()) <= 1e-3, errs


# ------------------------------------------------------------------------------------------------
# the fp16 mode on a second weight / input distribution ("trained-like": heavier tails, non-unit LayerNorm gains)
# ------------------------------------------------------------------------------------------------
def trained_like(cfg, sd, inp, seed=11, row_scale=(2.0, 4.0), gain=(0.5, 3.0), shift=0.2, sigma=0.5):
    """rows of in_proj / linear1 scaled x2-4, LayerNorm gains in [0.5, 3] with non-zero shifts, log-normal feature magnitudes:
    what a trained checkpoint looks like next to the U(+-1/sqrt(fan_in)) synthetic init"""
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.clone() for k, v in sd.items()}
    for k in sd:
        if k.endswith("self_attn.in_proj_weight") or k.endswith("linear1.weight"):
            sd[k] *= (row_scale[0] + (row_scale[1] - row_scale[0]) * torch.rand(sd[k].shape[0], 1, generator=g))
        elif ".norm" in k and k.endswith(".weight") or k.endswith("embedder.3.weight") or k == "time_mlp.6.weight":
            sd[k] = gain[0] + (gain[1] - gain[0]) * torch.rand(sd[k].shape, generator=g)
        elif ".norm" in k and k.endswith(".bias"):
            sd[k] = shift * torch.randn(sd[k].shape, generator=g)
    inp = {k: v.clone() for k, v in inp.items()}
    for k in ("visual", "audio"):
        if inp[k].dim() == 3:
            inp[k] *= torch.exp(sigma * torch.randn(inp[k].shape, generator=g))
    return sd, inp


MODERATE = dict(row_scale=(1.5, 3.0), gain=(0.5, 2.0), shift=0.2, sigma=0.25)
HARSH = dict(row_scale=(2.0, 4.0), gain=(0.5, 3.0), shift=0.2, sigma=0.5)


@pytest.mark.parametrize("train", [False, True])
def test_c2a_fp16_trained_like_weights(train):
    """fp16 mode, C2a B = 2, on a trained-like distribution (in_proj / linear1 rows x1.5-3, LayerNorm gains 0.5-2 with shifts,
    log-normal feature magnitudes; logits up to ~3).  Evaluation mode: every logit within 1e-3 of max(1, largest |logit|) of the
    fp32 oracle.  Training mode (same masks; dropout 0.5 / 0.5 doubles twice what survives): within 2.5e-3, and in both modes
    below the error of the reference's own GPU arithmetic on the same inputs (fp16 autocast: EVERY Linear on fp16 operands,
    scripts/train.py:197 - the oracle with rd = fp16).  Gradients: cos >= 0.9995."""
    cfg = named_config("C2a")
    B, nv, na = 2, 15, 10
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=2, dtype=torch.float32)
    sd, inp = trained_like(cfg, sd, inp, **MODERATE)
    with torch.no_grad():
        o_eval = H.named_outputs(*O.forward(sd, cfg, inp["visual"], inp["audio"], inp["times"], nv, na))
    R = H.cotangents(cfg, B, nv, na, o_eval, seed=2, dtype=torch.float32)
    m = build(cfg, "fp16", sd)
    masks = None
    if train:
        res = run_train(m, inp, nv, na, R)
        masks = site_masks(cfg, res["seed"], B, res["S"], inp)
        o, g, _ = oracle_train(cfg, sd, inp, nv, na, R, masks)
    else:
        from tests.test_gpu_parity import run_model, oracle_run
        res = run_model(m, inp, nv, na, True, R)
        o, _, g, _ = oracle_run(cfg, sd, inp, nv, na, R, torch.float32)
    with torch.no_grad():
        rec = H.named_outputs(*O.forward(sd, cfg, inp["visual"], inp["audio"], inp["times"], nv, na, masks=masks, rd=torch.float16))
    heads = [k for k in res["outs"] if k != "feats"]
    worst = max(maxerr(res["outs"][k], o[k]) for k in heads)
    recipe = max(maxerr(rec[k], o[k]) for k in heads)
    scale = max(amax(o[k]) for k in heads)
    wc = min(grad_agreement(res["grads"][k], v)[0] for k, v in g.items() if not k.startswith("drloc_mlp"))
    print("C2a fp16 trained-like weights (train=%s): worst |dlogit| %.3g at |logit|max %.2f -> %.3g relative (reference fp16-autocast "
          "arithmetic: %.3g); min gradient cos %.6f" % (train, worst, scale, worst / max(1.0, scale), recipe, wc))
    assert worst / max(1.0, scale) <= (2.5e-3 if train else 1e-3), worst
    assert worst <= recipe
    assert wc >= 0.9995, wc


def test_c2a_ill_conditioned_weights_no_worse_than_reference_fp16_recipe():
    """A deliberately ill-conditioned distribution (rows x2-4 AND gains up to 3: saturated attention, a rounding of 2^-12 in
    layer 0 moves the logits by 6e-2 - tools/err_budget.py).  No 16-bit-operand evaluation reaches 1e-3 here, the reference's own
    GPU recipe (fp16 autocast: every Linear on fp16 operands, scripts/train.py:197) included; what is asserted: the fp16 mode's
    error is below that recipe's, the fp32 kernels stay within 1e-5 x the same amplification, bf16x3 (three-pass) within 2e-3."""
    cfg = named_config("C2a")
    B, nv, na = 2, 15, 10
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=2, dtype=torch.float32)
    sd, inp = trained_like(cfg, sd, inp, **HARSH)
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        ref = H.named_outputs(*O.forward(sd64, cfg, inp["visual"].double(), inp["audio"].double(), inp["times"].double(), nv, na))
        rec = H.named_outputs(*O.forward(sd, cfg, inp["visual"], inp["audio"], inp["times"], nv, na, rd=torch.float16))
    heads = [k for k in ref if k != "feats"]
    e_recipe = max(maxerr(rec[k], ref[k]) for k in heads)
    errs = {}
    from tests.test_gpu_parity import run_model
    for prec in ("fp16", "fp32", "bf16x3"):
        m = build(cfg, prec, sd)
        res = run_model(m, inp, nv, na, False)
        errs[prec] = max(maxerr(res["outs"][k], ref[k]) for k in heads)
    print("ill-conditioned C2a: reference fp16-autocast arithmetic %.3g; HIP fp16 %.3g, bf16x3 %.3g, fp32 %.3g (|logit|max %.2f)"
          % (e_recipe, errs["fp16"], errs["bf16x3"], errs["fp32"], max(amax(ref[k]) for k in heads)))
    assert errs["fp16"] <= e_recipe
    assert errs["fp32"] <= 5e-4      # (fp32 rounding, 6e-8, times the same ~250x amplification)
    assert errs["bf16x3"] <= 2e-3


# ------------------------------------------------------------------------------------------------
# detection in .train(): the step bench.py's c4_train block times (drawn queries, on-device labelling, all dropout sites)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec,B", [("fp32", 2), ("fp16", 2)])
def test_detection_train_mode_vs_oracle_with_masks(prec, B):
    """det models/tim.py:272-345 (forward_train): the query set is drawn from the training pyramid, every query labelled
    against the ground truth, the encoder runs with all dropout sites on.  The model returns the queries it drew; the oracle
    gets them and the step's masks: classification / regression outputs <= 1e-5 (fp32 kernels) / 1e-3 (fp16), the labelling
    bit-identical to the oracle's (itself pinned to the reference's label_queries), every parameter gradient of a
    fixed-cotangent loss within 2e-4 (fp32) / cos >= 0.998 (fp16, toy width)."""
    im, dm, nc, tag = H.DET_CASES[2]          # audio_visual / audio_visual, verb + noun + action heads and the audio head
    cfg = H.tiny_cfg("detection", im, dm, tag == "vn", num_class=nc)
    cfg.feat_drop, cfg.seq_drop, cfg.enc_dropout = 0.3, 0.25, 0.2
    sd, inp = H.synth_torch(cfg, B, 0, 0, seed=3, dtype=torch.float32)
    m = build(cfg, prec, sd).train()
    g = torch.Generator().manual_seed(21)
    ngt = 4
    segs = lambda: torch.sort(torch.rand(B, ngt, 2, generator=g), dim=-1)[0]
    vc = nc[0]
    ri = lambda hi: torch.randint(0, hi, (B, ngt), generator=g)
    target = {"v_gt_segments": segs(), "a_gt_segments": segs(), "verb": ri(vc[0]), "noun": ri(vc[1]), "action": ri(vc[2]),
              "class_id": ri(nc[1])}
    tdev = {k: v.to(DEV) for k, v in target.items()}
    vis = inp["visual"].to(DEV).requires_grad_(True)
    aud = inp["audio"].to(DEV).requires_grad_(True)
    m.rt.step = 100
    (cls, reg, feats), offsets, labels, queries, ious = m([vis, aud], "encoder", inp["times"].to(DEV), tdev, label_queries=True)
    seed = m.rt.last_seed
    node = cls[2].grad_fn
    S = node.dims[3]
    nq = m.num_queries
    vq, aq = queries[0].detach().cpu().view(B, nq, 2), queries[1].detach().cpu().view(B, nq, 2)
    assert not torch.equal(vq[0], m.inference_queries[0])          # really the drawn training set
    # labelling: bit-identical to the oracle on the drawn queries
    for mod, q, sg, lab, ncs in (("visual", vq, target["v_gt_segments"], torch.stack([target["verb"], target["noun"], target["action"]], -1), list(vc)),
                                 ("audio", aq, target["a_gt_segments"], target["class_id"].unsqueeze(-1), [nc[1]])):
        tg, mats, best = O.label_queries(q, sg, lab, m.iou_threshold, m.label_smoothing, ncs)
        i = 0 if mod == "visual" else 1
        assert torch.equal(offsets[i].cpu(), tg), mod
        assert torch.equal(ious[i].cpu(), best), mod
        got = labels[0] if mod == "visual" else [labels[1]]
        for a, b in zip(got, mats):
            assert torch.equal(a.cpu(), b), mod
    outs = H.named_outputs(cls, feats, reg)
    gR = torch.Generator().manual_seed(5)
    R = {k: torch.randn(v.shape, generator=gR) * 0.1 for k, v in outs.items()}
    sum((outs[k] * R[k].to(DEV)).sum() for k in outs).backward()
    torch.cuda.synchronize()
    # the oracle on the same queries under the same masks
    masks = site_masks(cfg, seed, B, S, inp)
    sdo = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    times = torch.cat([inp["times"], vq, aq], 1).double()
    vo, ao = inp["visual"].double().clone().requires_grad_(True), inp["audio"].double().clone().requires_grad_(True)
    ocls, ofeats, oreg = O.forward(sdo, cfg, vo, ao, times, nq, nq, masks=masks)
    oo = H.named_outputs(ocls, ofeats, oreg)
    assert set(oo) == set(outs)
    sum((oo[k] * R[k].double()).sum() for k in oo).backward()
    tol = {"fp32": 1e-5, "fp16": 1e-3}[prec]
    for k, v in outs.items():
        e = maxerr(v.detach().cpu(), oo[k].detach()) / max(1.0, amax(oo[k].detach()))
        assert e <= tol, (prec, k, e)
    wc, wr = 1.0, 0.0
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
    for k, v in sdo.items():
        if v.grad is None or k.startswith("drloc_mlp"):
            continue
        cos, rel = grad_agreement(grads[k], v.grad)
        wc, wr = min(wc, cos), max(wr, rel)
        # (fp16 on a 32-wide toy model: a handful of ReLU decisions of the regression MLPs flip under 11-bit operands)
        assert cos >= (0.999999 if prec == "fp32" else 0.998), (prec, k, cos)
        assert rel <= (2e-4 if prec == "fp32" else 0.1), (prec, k, rel)
    assert relerr(vis.grad.cpu(), vo.grad) <= (2e-4 if prec == "fp32" else 0.1)
    print("detection train mode %s: gradients min cos %.6f, max rel err %.3g" % (prec, wc, wr))
