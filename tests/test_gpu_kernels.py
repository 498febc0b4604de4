"""Kernel-level parity through the C ABI (include/timhip.h) on a real MI355X.

Each HIP kernel is compared with the corresponding oracle function
(oracle/tim_oracle.py) or a plain torch-CPU fp32/fp64 statement of the same op on
identical seeded inputs.  Tolerances: fp32 kernels 1e-5 relative to the output
scale; bf16 kernels are compared with the same op evaluated on bf16-rounded
operands (fp32 accumulate), tolerance 1e-3 relative to the output scale.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tim_oracle as O  # noqa: E402
from tim_amd import _lib as L  # noqa: E402
from tim_amd.functional import Runtime, _ru  # noqa: E402

DEV = "cuda:0"
PRECS = ["fp32", "bf16x3", "bf16", "fp16"]
H16 = ["bf16", "fp16"]
HALF_ULP = {"bf16": 2 ** -8, "fp16": 2 ** -11}   # relative rounding step of a stored 16-bit result


def st():
    return torch.cuda.current_stream().cuda_stream


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def to_op(rt, x, ld=None):
    """fp32 CPU [R,C] -> device operand buffer [R, ru(C)] (zero padded) and its value as fp32 CPU"""
    R, Cc = x.shape
    buf = torch.zeros((R, ld or _ru(Cc)), dtype=rt.op_dtype, device=DEV)
    buf[:, :Cc] = x.to(DEV).to(rt.op_dtype)
    return buf, buf[:, :Cc].float().cpu()


def tol(prec, ref):
    s = max(1.0, float(ref.detach().abs().max()))
    return {"fp32": 1e-5, "bf16x3": 5e-5, "bf16": 1e-3, "fp16": 1.5e-4}[prec] * s


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 192), (9, 3806, 128), (333, 64, 2304), (1, 5, 24),
                                   (9925, 1000, 128), (4000, 2056, 64)])  # the last two select the 160-row tile
def test_gemm_store(prec, M, N, K):
    rt = Runtime(prec)
    A, Ar = to_op(rt, rnd(M, K, seed=1))
    B, Br = to_op(rt, rnd(N, K, seed=2, scale=K ** -0.5))
    bias = rnd(N, seed=3).to(DEV)
    ref = Ar.double() @ Br.double().t() + bias.cpu().double()
    # fp32 output with an unaligned leading dimension (N may be odd)
    out = torch.full((M, N), float("nan"), device=DEV)
    rt.gemm(L.EPI_STORE_F32, A, B, M, N, K, out, N, bias=bias)
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= tol(prec, ref), err
    # operand-dtype output into a padded buffer; pad columns must stay untouched
    outT = torch.zeros((M, _ru(N)), dtype=rt.op_dtype, device=DEV)
    rt.gemm(L.EPI_RELU_T, A, B, M, N, K, outT, outT.shape[1], bias=bias)
    torch.cuda.synchronize()
    got = outT.float().cpu()
    assert (got[:, N:] == 0).all()
    lim = tol(prec, ref) + HALF_ULP.get(prec, 0.0) * float(ref.abs().max())
    assert (got[:, :N].double() - ref.clamp(min=0)).abs().max().item() <= lim


@pytest.mark.parametrize("prec", PRECS)
def test_gemm_transpose_detecting(prec):
    """A = I-like with an asymmetric B: catches swapped row/col output mappings."""
    rt = Runtime(prec)
    M = N = K = 128
    a = torch.zeros(M, K)
    a[torch.arange(M), (torch.arange(M) * 7 + 3) % K] = 1.0
    b = (torch.arange(N).float()[:, None] * 0.01 + torch.arange(K).float()[None, :] * 0.5) / 64.0
    A, Ar = to_op(rt, a)
    B, Br = to_op(rt, b)
    out = torch.empty((M, N), device=DEV)
    rt.gemm(L.EPI_STORE_F32, A, B, M, N, K, out, N)
    torch.cuda.synchronize()
    ref = Ar @ Br.t()
    assert (out.cpu() - ref).abs().max().item() <= 1e-5


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,N,K", [(310, 256, 128), (4000, 2056, 128)])  # 128-row and 160-row bf16 tiles
def test_gemm_fused_epilogues(prec, M, N, K):
    rt = Runtime(prec)
    A, Ar = to_op(rt, rnd(M, K, seed=1))
    B, Br = to_op(rt, rnd(N, K, seed=2, scale=K ** -0.5))
    bias = rnd(N, seed=3).to(DEV)
    res = rnd(M, N, seed=4).to(DEV)
    lin = Ar.double() @ Br.double().t() + bias.cpu().double()
    # residual, no dropout
    out = torch.empty((M, N), device=DEV)
    rt.gemm(L.EPI_DROP_RES_F32, A, B, M, N, K, out, N, bias=bias, res=res, ldres=N)
    torch.cuda.synchronize()
    ref = res.cpu().double() + lin
    assert (out.cpu().double() - ref).abs().max().item() <= tol(prec, ref)
    # residual with dropout: the kept set must equal timhip_dropout_mask for the same (seed, site)
    p, seed, site = 0.3, 1234567, 42
    mask = torch.empty((M, N), dtype=torch.uint8, device=DEV)
    L.call("timhip_dropout_mask", seed, site, p, M, N, L.ptr(mask), st())
    rt.gemm(L.EPI_DROP_RES_F32, A, B, M, N, K, out, N, bias=bias, res=res, ldres=N, p_drop=p, seed=seed, site=site)
    torch.cuda.synchronize()
    mk = mask.cpu().double()
    assert abs(mk.mean().item() - (1 - p)) < 0.01
    ref = res.cpu().double() + lin * mk / (1 - p)
    assert (out.cpu().double() - ref).abs().max().item() <= tol(prec, ref)
    # GELU + dropout, two outputs
    u = torch.zeros((M, N), dtype=rt.op_dtype, device=DEV)
    h = torch.zeros((M, N), dtype=rt.op_dtype, device=DEV)
    rt.gemm(L.EPI_GELU_DROP_T2, A, B, M, N, K, h, N, out1=u, ld1=N, bias=bias, p_drop=p, seed=seed, site=site)
    torch.cuda.synchronize()
    rnd_tol = HALF_ULP.get(prec, 0.0) * float(lin.abs().max()) * 2
    assert (u.float().cpu().double() - lin).abs().max().item() <= tol(prec, lin) + rnd_tol
    href = O._gelu(lin) * mk / (1 - p)
    assert (h.float().cpu().double() - href).abs().max().item() <= tol(prec, href) + rnd_tol
    # gelu' * mask epilogue (dgrad through the FFN)
    g = torch.zeros((M, N), dtype=rt.op_dtype, device=DEV)
    rt.gemm(L.EPI_DGELU_T, A, B, M, N, K, g, N, aux=u, ldaux=N, p_drop=p, seed=seed, site=site)
    torch.cuda.synchronize()
    uu = u.float().cpu().double().requires_grad_(True)
    O._gelu(uu).sum().backward()
    gref = (Ar.double() @ Br.double().t()) * mk / (1 - p) * uu.grad
    assert (g.float().cpu().double() - gref).abs().max().item() <= tol(prec, gref) + rnd_tol
    # split-K accumulate
    acc = torch.ones((M, N), device=DEV)
    rt.gemm(L.EPI_ATOMIC_F32, A, B, M, N, K, acc, N, splitk=2)
    torch.cuda.synchronize()
    ref = 1.0 + Ar.double() @ Br.double().t()
    assert (acc.cpu().double() - ref).abs().max().item() <= tol(prec, ref)


@pytest.mark.parametrize("prec", PRECS)
def test_weight_operand_copies(prec):
    """Runtime.weight: plain and transposed operand copies of fp32 masters (batched refresh, one launch for all
    stale weights), zero padded to multiples of 64; ragged shapes take the scalar-load path."""
    rt = Runtime(prec)
    shapes = [(512, 2), (97, 1024), (3806, 1024), (300, 70), (1, 512), (130, 131), (2048, 1024)]
    ws = [torch.nn.Parameter(rnd(r, c, seed=40 + i).to(DEV)) for i, (r, c) in enumerate(shapes)]
    for w in ws:  # register; the first call casts one weight, later stale ones are refreshed together
        rt.weight(w)
    rt.invalidate_weights()
    for w, (r, c) in zip(ws, shapes):
        plain, tr = rt.weight(w), rt.weight(w, True)
        torch.cuda.synchronize()
        ref = w.detach().to(rt.op_dtype).float().cpu()
        assert plain.shape == (r, _ru(c)) and tr.shape == (c, _ru(r))
        assert torch.equal(plain.float().cpu()[:, :c], ref)
        assert torch.equal(tr.float().cpu()[:, :r], ref.t())
        assert (plain.float().cpu()[:, c:] == 0).all() and (tr.float().cpu()[:, r:] == 0).all()
    # an in-place update (optimizer step) makes the copies stale again
    with torch.no_grad():
        ws[1].mul_(2.0)
    torch.cuda.synchronize()
    got = rt.weight(ws[1]).float().cpu()[:, :1024]
    assert torch.equal(got, ws[1].detach().to(rt.op_dtype).float().cpu())


@pytest.mark.parametrize("prec", PRECS)
def test_wgrad_and_colsum(prec):
    rt = Runtime(prec)
    M, N, K = 523, 200, 72
    dY, dYr = to_op(rt, rnd(M, N, seed=5))
    X, Xr = to_op(rt, rnd(M, K, seed=6))
    dW = torch.ones((N, K), device=DEV)
    db = torch.ones((N,), device=DEV)
    rt.wgrad(dY, N, X, K, M, dW, db)
    torch.cuda.synchronize()
    ref = 1.0 + dYr.double().t() @ Xr.double()
    assert (dW.cpu().double() - ref).abs().max().item() <= tol(prec, ref) * 4
    refb = 1.0 + dYr.double().sum(0)
    assert (db.cpu().double() - refb).abs().max().item() <= tol(prec, refb) * 4


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,N,K,p", [(333, 256, 128, 0.0), (9925, 1024, 64, 0.25), (70, 136, 192, 0.25)])
def test_gemm_residual_is_layernorm(prec, M, N, K, p):
    """DROP_RES_F32 with TimEpi.ln_* (residual = LayerNorm(res), normalised by the epilogue) == LayerNorm kernel first, then
    the plain residual epilogue"""
    rt = Runtime(prec)
    A, _ = to_op(rt, rnd(M, K, seed=1))
    B, _ = to_op(rt, rnd(N, K, seed=2, scale=K ** -0.5))
    y = (rnd(M, N, seed=3) * 2 + 0.5).to(DEV)
    g = (1 + 0.1 * rnd(N, seed=4)).to(DEV)
    b = (0.1 * rnd(N, seed=5)).to(DEV)
    bias = (0.1 * rnd(N, seed=6)).to(DEV)
    xf = torch.empty((M, N), device=DEV)
    stats = torch.empty((M, 2), device=DEV)
    rt.ln_fwd(y, M, N, 0, g, b, xf=xf, ldx=N, stats=stats)
    ref = torch.empty((M, N), device=DEV)
    out = torch.empty((M, N), device=DEV)
    kw = dict(bias=bias, ldres=N, p_drop=p, seed=77, site=5)
    rt.gemm(L.EPI_DROP_RES_F32, A, B, M, N, K, ref, N, res=xf, **kw)
    rt.gemm(L.EPI_DROP_RES_F32, A, B, M, N, K, out, N, res=y, ln=(stats, g, b), **kw)
    torch.cuda.synchronize()
    assert (out - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("prec", H16)
@pytest.mark.parametrize("epi", ["store_f32", "add_f32"])
def test_gemm_group(epi, prec):
    """several independent small problems as one grouped launch == one launch each (the classification heads' shapes: ragged
    N, different M, ragged K)"""
    rt = Runtime(prec)
    shapes = [(960, 97, 1024), (960, 300, 1024), (960, 3806, 1024), (640, 44, 1024), (70, 136, 192), (1, 5, 24)]
    items, refs = [], []
    for i, (M, N, K) in enumerate(shapes):
        A, _ = to_op(rt, rnd(M, K, seed=10 + i))
        B, _ = to_op(rt, rnd(N, K, seed=30 + i, scale=K ** -0.5))
        bias = (0.1 * rnd(N, seed=50 + i)).to(DEV)
        res = rnd(M, N, seed=70 + i).to(DEV)
        out = torch.full((M, N), 7.0, device=DEV)
        ref = torch.full((M, N), 7.0, device=DEV)
        if epi == "store_f32":
            items.append(dict(A=A, B=B, M=M, N=N, K=K, out0=out, ld0=N, bias=bias))
            rt.gemm(L.EPI_STORE_F32, A, B, M, N, K, ref, N, bias=bias)
        else:
            items.append(dict(A=A, B=B, M=M, N=N, K=K, out0=out, ld0=N, res=res, ldres=N))
            rt.gemm(L.EPI_ADD_F32, A, B, M, N, K, ref, N, res=res, ldres=N)
        refs.append(ref)
    rt.gemm_many(L.EPI_STORE_F32 if epi == "store_f32" else L.EPI_ADD_F32, items)
    torch.cuda.synchronize()
    for it, ref in zip(items, refs):
        assert torch.equal(it["out0"], ref)


@pytest.mark.parametrize("prec", H16)
@pytest.mark.parametrize("epi", ["gelu_drop", "dgelu"])
@pytest.mark.parametrize("M,N", [(9925, 2048), (130, 72), (64, 256)])
def test_gemm_dropout_keep_bits(epi, M, N, prec):
    """the FFN epilogues with the keep-bits drawn ahead of time (TimEpi.mask: what LayerNorm-1 writes inside the layer) ==
    the same epilogues drawing Philox themselves; the bits here come from the timhip_dropout_mask test hook"""
    rt = Runtime(prec)
    K, p, seed, site = 128, 0.3, 1234, 16 + 8 + 3
    A, _ = to_op(rt, rnd(M, K, seed=1))
    B, _ = to_op(rt, rnd(N, K, seed=2, scale=K ** -0.5))
    bias = (0.1 * rnd(N, seed=6)).to(DEV)
    ld = _ru(N)
    keep = torch.empty((M, (N + 3) // 4 * 4), dtype=torch.uint8, device=DEV)
    L.call("timhip_dropout_mask", seed, site, p, M, N, L.ptr(keep), st())
    ldm = (N + 7) // 8
    padded = torch.zeros((M, ldm * 8), dtype=torch.uint8, device=DEV)
    padded[:, :N] = keep[:, :N]
    bits = (padded.view(M, ldm, 8).to(torch.int32) << torch.arange(8, device=DEV, dtype=torch.int32)).sum(-1).to(torch.uint8)
    outs = []
    for mask in (None, bits):
        o0 = torch.zeros((M, ld), dtype=rt.op_dtype, device=DEV)
        o1 = torch.zeros((M, ld), dtype=rt.op_dtype, device=DEV)
        aux = (rnd(M, ld, seed=9)).to(DEV).to(rt.op_dtype)
        kw = dict(p_drop=p, seed=seed, site=site, mask=mask, ldmask=ldm)
        if epi == "gelu_drop":
            rt.gemm(L.EPI_GELU_DROP_T2, A, B, M, N, K, o0, ld, out1=o1, ld1=ld, bias=bias, **kw)
        else:
            rt.gemm(L.EPI_DGELU_T, A, B, M, N, K, o0, ld, aux=aux, ldaux=ld, **kw)
        torch.cuda.synchronize()
        outs.append((o0.clone(), o1.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert (outs[0][0][:, :N] == 0).float().mean().item() > p * 0.8       # dropout really happened


@pytest.mark.parametrize("prec", H16)
@pytest.mark.parametrize("accumulate", [True, False])
@pytest.mark.parametrize("M,shapes", [
    (523, [(200, 72), (64, 136), (130, 128)]),                       # ragged tiles, few tiles: the contraction is split
    (155 * 8, [(256, 512), (512, 256), (256, 256), (768, 256)]),      # an encoder layer (E 256, FF 512): 48 tiles, split
    (9920, [(1024, 2048), (2048, 1024), (1024, 1024), (3072, 1024)]),  # C2a: 512 tiles, no split, direct writes
    (4100, [(1024, 2048), (2048, 1024), (1024, 1024), (3072, 1024)]),  # same, ragged last step of 64 rows
    (2100, [(2000, 1000), (1000, 2000), (3000, 1000)]),                # one-block-per-CU kernels with ragged n / k tiles and rows
])
@pytest.mark.parametrize("loaders", [True, False])
def test_wgrad_group(M, shapes, accumulate, prec, loaders, knobs):
    """several Linear weight gradients sharing M in one launch == the per-layer reference; biases optional.
    loaders: the one-block-per-CU kernel with four DMA loader waves (default) / its 8-wave form (TIMHIP_WGRAD_LD=0)"""
    if not loaders:
        if M < 2048:
            pytest.skip("small groups do not take the one-block-per-CU kernels")
        knobs(TIMHIP_WGRAD_LD="0")
    rt = Runtime(prec)
    items, refs = [], []
    for i, (N, K) in enumerate(shapes):
        dY, dYr = to_op(rt, rnd(M, N, seed=20 + i) * 0.25)
        X, Xr = to_op(rt, rnd(M, K, seed=40 + i) * 0.25)
        dW = torch.full((N, K), 0.5, device=DEV)
        db = None if i == 1 else torch.full((N,), 0.5, device=DEV)
        items.append((dY, N, X, K, dW, db))
        base = 0.5 if accumulate else 0.0
        refs.append((base + dYr.to(DEV).double().t() @ Xr.to(DEV).double(), base + dYr.double().sum(0)))
    rt.wgrad_group(items, M, accumulate=accumulate)
    torch.cuda.synchronize()
    for (dY, N, X, K, dW, db), (rw, rb) in zip(items, refs):
        assert (dW.double() - rw).abs().max().item() <= 2e-5 * max(1.0, rw.abs().max().item()) * (M / 512) ** 0.5 + 1e-4
        if db is not None:
            assert (db.cpu().double() - rb).abs().max().item() <= 2e-5 * max(1.0, rb.abs().max().item()) + 1e-4


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("M", [2048, 9920, 7984, 2051])   # (7984 = C4's 16 x 499 rows, 2051: ragged last stages of 48 / 3 rows)
@pytest.mark.parametrize("phases", [4, 2])
def test_wgrad_group_eight_phase(M, accumulate, prec, phases, knobs):
    """round 6: the weight gradients of TWO encoder layers as one round of 256 x 256 eight-phase tiles (wgrad_p8_kernel) == the
    per-product reference, == the one-block-per-CU kernel's results bit for bit in its biases' tolerance; TIMHIP_WGRAD_P8=0
    sends the same group through the 128 x 256 kernel"""
    knobs(TIMHIP_WGRAD_P8_PH=str(phases))   # (four 16-MFMA phases per contraction step, or two 32-MFMA ones)
    rt = Runtime(prec)
    shapes = [(1024, 2048), (2048, 1024), (1024, 1024), (3072, 1024)] * 2
    items, refs = [], []
    for i, (N, K) in enumerate(shapes):
        dY, dYr = to_op(rt, rnd(M, N, seed=120 + i) * 0.25)
        X, Xr = to_op(rt, rnd(M, K, seed=140 + i) * 0.25)
        dW = torch.full((N, K), 0.5, device=DEV)
        db = None if i == 5 else torch.full((N,), 0.5, device=DEV)
        items.append((dY, N, X, K, dW, db))
        base = 0.5 if accumulate else 0.0
        refs.append((base + dYr.to(DEV).double().t() @ Xr.to(DEV).double(), base + dYr.double().sum(0)))
    rt.wgrad_group(items, M, accumulate=accumulate)
    torch.cuda.synchronize()
    got = [(dW.clone(), None if db is None else db.clone()) for (_, _, _, _, dW, db) in items]
    for (dW, db), (rw, rb) in zip(got, refs):
        assert (dW.double() - rw).abs().max().item() <= 2e-5 * max(1.0, rw.abs().max().item()) * (M / 512) ** 0.5 + 1e-4
        if db is not None:
            assert (db.cpu().double() - rb).abs().max().item() <= 2e-5 * max(1.0, rb.abs().max().item()) + 1e-4
    # the same group through the 128 x 256 kernel: same products, another summation order inside the MFMA chain only
    knobs(TIMHIP_WGRAD_P8="0", TIMHIP_WGRAD_P8_PH=str(phases))
    for (_, _, _, _, dW, db) in items:
        dW.fill_(0.5)
        if db is not None:
            db.fill_(0.5)
    rt.wgrad_group(items, M, accumulate=accumulate)
    torch.cuda.synchronize()
    for (dW0, db0), (_, _, _, _, dW, db) in zip(got, items):
        assert (dW0 - dW).abs().max().item() <= 1e-4 * max(1.0, dW.abs().max().item())
        if db is not None:
            assert (db0 - db).abs().max().item() <= 1e-4 * max(1.0, db.abs().max().item())


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("cols,act", [(1024, 0), (512, 2), (64, 1), (32, 2), (256, 0)])
def test_layernorm_fwd_bwd(prec, cols, act):
    rt = Runtime(prec)
    rows = 77
    y = rnd(rows, cols, seed=7)
    w = 1 + 0.1 * rnd(cols, seed=8)
    b = 0.1 * rnd(cols, seed=9)
    dxo = rnd(rows, cols, seed=10)
    yd, wd, bd, dxd = [t.to(DEV) for t in (y, w, b, dxo)]
    xf = torch.empty((rows, cols), device=DEV)
    xt = torch.zeros((rows, _ru(cols)), dtype=rt.op_dtype, device=DEV)
    stats = torch.empty((rows, 2), device=DEV)
    rt.ln_fwd(yd, rows, cols, act, wd, bd, xf=xf, ldx=cols, xt=xt, ldt=xt.shape[1], stats=stats)
    torch.cuda.synchronize()
    y64 = y.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    a = {0: lambda t: t, 1: torch.relu, 2: O._gelu}[act](y64)
    ref = O._ln(a, w64, b64)
    assert (xf.cpu().double() - ref.detach()).abs().max().item() <= 2e-5
    assert (xt[:, :cols].float().cpu().double() - ref.detach()).abs().max().item() <= {"bf16": 0.05, "fp16": 0.01}.get(prec, 2e-5)
    (ref * dxo.double()).sum().backward()
    dyf = torch.empty((rows, cols), device=DEV)
    dg = torch.zeros(cols, device=DEV)
    dbeta = torch.zeros(cols, device=DEV)
    rt.ln_bwd(dxd, yd, stats, rows, cols, act, wd, dyf=dyf, dgamma=dg, dbeta=dbeta)
    torch.cuda.synchronize()
    assert (dyf.cpu().double() - y64.grad).abs().max().item() <= 1e-4 * max(1.0, y64.grad.abs().max().item())
    assert (dg.cpu().double() - w64.grad).abs().max().item() <= 1e-4 * max(1.0, w64.grad.abs().max().item())
    assert (dbeta.cpu().double() - b64.grad).abs().max().item() <= 1e-4 * max(1.0, b64.grad.abs().max().item())


def _attn_case(prec, B, S, F, H, Dh, p=0.0, seed=11):
    rt = Runtime(prec)
    E = H * Dh
    qkv, qkvr = to_op(rt, rnd(B * S, 3 * E, seed=seed, scale=1.0), ld=3 * E)
    do, dor = to_op(rt, rnd(B * S, E, seed=seed + 1), ld=E)
    desc = L.TimDesc(B, S, F, E // 2, E, H, 4 * E, rt.prec, p, 99, 1, 0)
    o = torch.zeros((B * S, E), dtype=rt.op_dtype, device=DEV)
    lse = torch.empty((B, H, S), device=DEV)
    L.call("timhip_attention_fwd", C.byref(desc), L.ptr(qkv), L.ptr(o), L.ptr(lse), st())
    torch.cuda.synchronize()
    mask = None
    if p > 0:
        LP = (F + 1 + 7) // 8 * 8     # row pitch of the probability dropout stream
        mk = torch.empty((B * H * S, LP), dtype=torch.uint8, device=DEV)
        L.call("timhip_dropout_mask", 99, 16 + 8 * 1 + 0, p, B * H * S, LP, L.ptr(mk), st())
        torch.cuda.synchronize()
        mask = mk.cpu().view(B, H, S, LP)[..., :F + 1].double()
    x = qkvr.double().view(B, S, 3, H, Dh).requires_grad_(True)
    q, k, v = [x[:, :, i].transpose(1, 2) for i in range(3)]
    ref = O.attention_structured(q, k, v, F, mask, p).transpose(1, 2).reshape(B * S, E)
    rnd_tol = HALF_ULP.get(prec, 0.0) * float(ref.abs().max())
    err = (o.float().cpu().double() - ref.detach()).abs().max().item()
    assert err <= tol(prec, ref) + rnd_tol, ("fwd", err)
    # backward
    (ref * dor.double()).sum().backward()
    gref = x.grad.reshape(B * S, 3 * E)
    dqkv = torch.zeros((B * S, 3 * E), dtype=rt.op_dtype, device=DEV)
    wsb = L.load().timhip_attention_bwd_workspace_bytes(C.byref(desc))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    L.call("timhip_attention_bwd", C.byref(desc), L.ptr(qkv), L.ptr(o), L.ptr(lse), L.ptr(do), L.ptr(dqkv),
           L.ptr(ws), wsb, st())
    torch.cuda.synchronize()
    rnd_tol = 2 * HALF_ULP.get(prec, 0.0) * float(gref.abs().max())
    err = (dqkv.float().cpu().double() - gref).abs().max().item()
    assert err <= 5 * tol(prec, gref) + rnd_tol, ("bwd", err)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,S,F,H,Dh", [(2, 22, 12, 2, 32), (2, 155, 100, 2, 128), (1, 80, 50, 1, 128),
                                        (1, 12, 12, 2, 64), (1, 205, 150, 1, 128), (1, 300, 100, 2, 128)])  # last: 10 row blocks on 8 waves
def test_attention_structured(prec, B, S, F, H, Dh):
    _attn_case(prec, B, S, F, H, Dh)


@pytest.mark.parametrize("prec", H16)
@pytest.mark.parametrize("B,S,F,H,Dh,p", [(2, 499, 100, 8, 128, 0.1), (1, 499, 100, 2, 128, 0.0), (3, 260, 50, 2, 64, 0.1)])
def test_attention_long_sequence_row_split(prec, B, S, F, H, Dh, p):
    """detection's sequence (S = 499: 100 feature tokens + 399 queries, 16 row blocks) with few (window, head) pairs: the row
    blocks of a pair are spread over several workgroups (attention_mfma.hip: attn_row_split) in the forward and in the row kernel
    of the backward - 4 parts of 4 row blocks at S = 499, ragged last part at S = 260 (9 row blocks); with attention dropout"""
    _attn_case(prec, B, S, F, H, Dh, p=p)


@pytest.mark.parametrize("prec", PRECS)
def test_attention_dropout(prec):
    _attn_case(prec, 2, 40, 24, 2, 64, p=0.25)


@pytest.mark.parametrize("prec", H16)
@pytest.mark.parametrize("fused", ["1", "1-one-wave-per-row-block", "0"])
@pytest.mark.parametrize("B,S,F,H,Dh,p", [(3, 155, 100, 2, 128, 0.1), (2, 125, 100, 1, 128, 0.0), (1, 160, 128, 2, 128, 0.1),
                                          (2, 192, 97, 1, 128, 0.0), (2, 129, 128, 1, 128, 0.1), (1, 98, 97, 2, 128, 0.1)])
def test_attention_backward_fused_and_two_kernel(prec, fused, B, S, F, H, Dh, p, knobs):
    """the production-shape backward in its three forms: rows + keys kernels with the dS / P~ scratch (TIMHIP_ATTN_FUSED=0), and the
    one-kernel form that keeps dS / P~ in LDS (128-wide heads, 97..128 feature keys, S <= 192) with phase 1 as the pipeline over
    row blocks (default) or one wave per row block (TIMHIP_ATTN_KS=0); with attention dropout.  Shapes: four, five and six row
    blocks, a last block of one row, no padding keys / 31 padding keys, one query row"""
    knobs(TIMHIP_ATTN_FUSED=fused[0], TIMHIP_ATTN_KS="0" if "one-wave" in fused else "1")
    _attn_case(prec, B, S, F, H, Dh, p=p)


def test_dropout_mask_statistics_and_determinism():
    for p in (0.1, 0.5):
        m1 = torch.empty((257, 1024), dtype=torch.uint8, device=DEV)
        m2 = torch.empty_like(m1)
        m3 = torch.empty_like(m1)
        L.call("timhip_dropout_mask", 7, 3, p, 257, 1024, L.ptr(m1), st())
        L.call("timhip_dropout_mask", 7, 3, p, 257, 1024, L.ptr(m2), st())
        L.call("timhip_dropout_mask", 8, 3, p, 257, 1024, L.ptr(m3), st())
        torch.cuda.synchronize()
        assert torch.equal(m1, m2)
        assert not torch.equal(m1, m3)
        keep = m1.float().mean().item()
        assert abs(keep - (1 - p)) < 4 * math.sqrt(p * (1 - p) / m1.numel()) + 1e-4
        # no row/column structure
        assert (m1.float().mean(0) - (1 - p)).abs().max().item() < 0.2
        assert abs(np.corrcoef(m1.cpu().numpy().ravel()[:-1], m1.cpu().numpy().ravel()[1:])[0, 1]) < 0.01


def test_grad_scale_kernel():
    """timhip_grad_scale: S = 2^floor(log2(target / max|cot|)) over several tensors, computed on the device; scratch left zero"""
    rt = Runtime("fp16")
    a = (rnd(1000, 37, seed=1) * 3e-4).to(DEV)
    b = (rnd(5, seed=2) * 1e-6).to(DEV)
    c = torch.zeros(0, device=DEV)
    for tensors in ([a, b, c], [b], [a[:7, :5].contiguous()]):
        gs = rt.grad_scale(tensors, DEV)
        torch.cuda.synchronize()
        amax = max(float(t.abs().max()) for t in tensors if t.numel())
        S = 2.0 ** math.floor(math.log2(rt.grad_scale_target / amax))
        assert gs.tolist() == [S, 1.0 / S] + [0.0] * 6, (gs.tolist(), S)   # {S, 1/S, scratch x2 left zero, non-finite flag, 0, 0, 0}
    gs = rt.grad_scale([torch.zeros(64, device=DEV)], DEV)
    torch.cuda.synchronize()
    assert gs.tolist()[:2] == [1.0, 1.0]
    assert Runtime("bf16").grad_scale([a], DEV) is None


def test_fp16_scaled_gradient_operands():
    """fp16 gradient operands stored times S: cast (x S) -> input-gradient GEMM (acc x 1/S) and weight-gradient GEMM (out x 1/S)
    reproduce the unscaled products of values that fp16 alone would flush to zero"""
    rt = Runtime("fp16")
    M, N, K = 300, 256, 128
    g = (rnd(M, N, seed=3) * 1e-7).to(DEV)          # below the fp16 subnormal range (6e-8) for most elements
    W = rnd(K, N, seed=4, scale=N ** -0.5)          # [K, N]: the transposed copy an input-gradient product reads
    X = rnd(M, K, seed=5)
    Wt, Wr = to_op(rt, W)
    Xt, Xr = to_op(rt, X)
    gs = rt.grad_scale([g], DEV)
    gT = torch.empty((M, N), dtype=torch.float16, device=DEV)
    L.call("timhip_cast_rows", rt.prec, L.ptr(g), M, N, N, L.ptr(gT), N, 0.0, 0, 0, L.ptr(gs), st())
    dx = torch.empty((M, K), device=DEV)
    rt.gemm(L.EPI_ADD_F32, gT, Wt, M, K, N, dx, K, acc_scale=L.ptr(gs) + 4)
    dW = torch.zeros((N, K), device=DEV)
    db = torch.zeros((N,), device=DEV)
    rt.wgrad(gT, N, Xt, K, M, dW, db, out_scale=L.ptr(gs) + 4)
    torch.cuda.synchronize()
    S = gs[0].item()
    gq = (g.cpu().double() * S).to(torch.float16).double() / S     # what the scaled fp16 operand holds
    ref_dx = gq @ Wr.double().t()
    ref_dW = gq.t() @ Xr.double()
    assert (dx.cpu().double() - ref_dx).abs().max().item() <= 1e-5 * ref_dx.abs().max().item()
    assert (dW.cpu().double() - ref_dW).abs().max().item() <= 1e-5 * ref_dW.abs().max().item()
    assert (db.cpu().double() - gq.sum(0)).abs().max().item() <= 1e-5 * gq.sum(0).abs().max().item()
    assert ref_dx.abs().max().item() > 0 and float(g.cpu().to(torch.float16).abs().max()) < 2e-6


_PP_SHAPES = [(9920, 1024, 1024), (9920, 3072, 1024), (9920, 1024, 2048), (9925, 1000, 192), (3000, 2056, 64),
              # M = 7984 / 8000 (detection B = 16 x 499 tokens, Perception Test): 50 panels of 160 rows would leave a fifth of
              # the CUs idle - these run the 128-row tile (63 panels: 252 tiles per 1024 columns), one tile per block and walked
              (7984, 1024, 1024), (7984, 3072, 1024), (8000, 2048, 192)]


def _pp_knobs(knobs, loaders):
    knobs(TIMHIP_GEMM_LD="0" if loaders is False else "1",
          TIMHIP_GEMM_PF="0" if loaders == "no_prefetch" else "4",
          TIMHIP_GEMM_PF_MR="0" if loaders == "no_prefetch" else "4",          # (multi-round shapes: off by default)
          TIMHIP_GEMM_PF_MODE="2" if loaders == "prefetch_all" else "1",
          TIMHIP_GEMM_LD1="1" if loaders == "one_barrier" else "0",
          # (the 744-tile shape runs as 248 blocks of three tiles by default, gemm_nt_ldp_kernel; TIMHIP_GEMM_LDP=0: one tile per block)
          TIMHIP_GEMM_LDP="0" if loaders == "one_tile_per_block" else "1",
          TIMHIP_GEMM_DG="0", TIMHIP_GEMM_PT="0")


@pytest.mark.parametrize("prec", H16)
@pytest.mark.parametrize("M,N,K", _PP_SHAPES)
@pytest.mark.parametrize("loaders", [False, True, "no_prefetch", "one_tile_per_block"])
def test_gemm_pingpong_kernel(prec, M, N, K, loaders, knobs):
    """gemm_pp.hip (one block per CU, 160 x 256 tiles, three-stage LDS ring) - the three kernels the product library carries:
    8 consumer + 4 DMA loader waves with the L2 prefetch of the tile's share of its XCD's lines, one tile per block
    (gemm_nt_ld_kernel) or a walk of 2-4 tiles (gemm_nt_ldp_kernel; the default for multi-round shapes, TIMHIP_GEMM_LDP=0: off),
    also without the prefetch; and the 8 waves that issue their DMA pieces themselves (gemm_nt_pp_kernel, TIMHIP_GEMM_LD=0, the
    fallback): every epilogue the kernel carries, on the encoder layer's shapes and on ragged edges (rows past M, columns
    past N, one / two / three contraction steps), against fp64"""
    _pp_knobs(knobs, loaders)
    _check_layer_gemm_epilogues(prec, M, N, K)


@pytest.mark.parametrize("tmw", ["4", "5"])
@pytest.mark.parametrize("M,N,K", [(7984, 1024, 1024), (7984, 2048, 192), (9925, 1000, 192), (8000, 3072, 128)])
def test_gemm_pingpong_row_tile_forced(M, N, K, tmw, knobs):
    """the 128-row (TMW = 4) and the 160-row (TMW = 5) tile of the loader-wave NT kernels FORCED (TIMHIP_GEMM_TMW) on row counts
    neither divides (round-4 advisor finding: the per-launch choice `pp_tmw` had no direct test): one tile per block and the
    walk, every epilogue, ragged last row panel and column tile, against fp64"""
    _pp_knobs(knobs, True)
    knobs(TIMHIP_GEMM_TMW=tmw)
    _check_layer_gemm_epilogues("fp16", M, N, K)


@pytest.mark.parametrize("prec", H16)
@pytest.mark.parametrize("tm", [8, 10])
@pytest.mark.parametrize("phases", [4, 2])
@pytest.mark.parametrize("M,N,K", [(9920, 3072, 1024), (9920, 2048, 1024), (9925, 2048, 192), (3000, 512, 128), (7984, 3072, 1024),
                                   (300, 256, 64 * 5), (13120, 1024, 2048)])
def test_gemm_eight_phase_kernel(prec, M, N, K, tm, phases, knobs):
    """gemm_nt_p8_kernel (round 6): 256 x 256 (TM = 8) and 320 x 256 (TM = 10) tiles on the eight-phase schedule - four quadrant
    phases per contraction step, half-tiles restaged one phase after their last fragment read, one counted vmcnt per step -
    FORCED (TIMHIP_GEMM_P8 = 8 / 10) on the layer's multi-round shapes, on row counts neither tile divides (ragged last panel: 9925,
    7984, 300 rows), one / two rounds of tiles, 2 .. 32 contraction steps: the three epilogues it carries (16-bit store + bias,
    GELU + dropout with two outputs, multiply by the saved factor) against fp64, the others through their usual kernels"""
    if phases == 4 and (M, N, K) not in ((9920, 3072, 1024), (9925, 2048, 192), (300, 256, 64 * 5)):
        pytest.skip("the four-phase loop (TIMHIP_GEMM_P8_PH=4, not the default any more) is kept tested on three shapes")
    knobs(TIMHIP_GEMM_P8=str(tm), TIMHIP_GEMM_P8_PH=str(phases))   # (phases per contraction step: four quadrants, or two row halves)
    for epi in (L.EPI_STORE_T, L.EPI_GELU_DROP_G2, L.EPI_MULAUX_T):
        assert L.load().timhip_gemm_p8_choice(epi, M, N, K) == tm     # really this kernel
    assert L.load().timhip_gemm_p8_choice(L.EPI_DROP_RES_F32, M, N, K) == 0
    _check_layer_gemm_epilogues(prec, M, N, K)


def test_gemm_eight_phase_choice(knobs):
    """which shapes take it by themselves (TIMHIP_GEMM_P8=1): C2a's three multi-round products - N = 3072 on 256 rows (468 tiles =
    1.83 rounds), N = 2048 on 320 rows (248 tiles = one round) - and nothing that fits one round of 160 x 256 tiles"""
    knobs(TIMHIP_GEMM_P8="1")
    ch = L.load().timhip_gemm_p8_choice
    assert ch(L.EPI_STORE_T, 9920, 3072, 1024) == 8
    assert ch(L.EPI_GELU_DROP_G2, 9920, 2048, 1024) == 10 and ch(L.EPI_MULAUX_T, 9920, 2048, 1024) == 0   # (measured: no gain with that epilogue)
    knobs(TIMHIP_GEMM_P8="2")
    assert ch(L.EPI_MULAUX_T, 9920, 2048, 1024) == 10
    knobs(TIMHIP_GEMM_P8="1")
    assert ch(L.EPI_STORE_T, 9920, 1024, 3072) == 0 and ch(L.EPI_STORE_T, 1240, 3072, 1024) == 0
    assert ch(L.EPI_STORE_T, 9920, 3072, 64) == 0          # one contraction step: the two-buffer prologue needs two
    knobs(TIMHIP_GEMM_P8="0")
    assert ch(L.EPI_STORE_T, 9920, 3072, 1024) == 0


@pytest.mark.tuning
@pytest.mark.parametrize("prec", H16)
@pytest.mark.parametrize("M,N,K", _PP_SHAPES)
@pytest.mark.parametrize("loaders", ["prefetch_all", "one_barrier"])
def test_gemm_pingpong_kernel_tuning_variants(prec, M, N, K, loaders, knobs):
    """TUNING=1 builds only: every line prefetched (measured: the scattered loads take the vector-memory path's time
    themselves), one barrier per contraction step (TIMHIP_GEMM_LD1=1: +0.5 % of the step)"""
    _pp_knobs(knobs, loaders)
    _check_layer_gemm_epilogues(prec, M, N, K)


@pytest.mark.tuning
@pytest.mark.parametrize("prec", H16)
@pytest.mark.parametrize("M,N,K", [(9920, 2048, 1024), (9920, 3072, 1024), (9920, 2048, 128), (9925, 3072, 192), (13120, 2048, 64 * 5)])
def test_gemm_persistent_tile_kernel(prec, M, N, K, knobs):
    """gemm_nt_pt_kernel (round 3; TUNING=1 builds only - measured equal to one tile per block): a block walks 2 or 3 (4 at
    M = 13120) consecutive 160 x 256 tiles, the next tile's first stages in flight during the epilogue (ring slot 2 is the
    transposition space); every epilogue, ragged last row panel, 2 .. 16 contraction steps"""
    knobs(TIMHIP_GEMM_DG="0", TIMHIP_GEMM_PT="1")
    _check_layer_gemm_epilogues(prec, M, N, K)


@pytest.mark.tuning
@pytest.mark.parametrize("prec", H16)
@pytest.mark.parametrize("M,N,K,offset", [(9920, 1024, 1024, 9), (9920, 2048, 1024, 9), (9920, 3072, 1024, 9), (9920, 1024, 2048, 9),
                                          (9920, 1024, 3072, 3), (8000, 1024, 128, 1), (13120, 1024, 192, 21)])
def test_gemm_dual_group_kernel(prec, M, N, K, offset, knobs):
    """gemm_nt_dg_kernel (round 3; TUNING=1 builds only - measured 13 % slower): two wave groups on different 160 x 128 tiles a
    few steps apart, persistent over tile pairs, 2-slot rings: every epilogue the encoder layers use, 1 / 2 / 3 pairs per
    block, 2 .. 48 contraction steps, several group offsets (the offset only changes WHEN things happen - results must not
    depend on it), against fp64"""
    knobs(TIMHIP_GEMM_DG="1", TIMHIP_GEMM_DG_OFFSET=str(offset))
    _check_layer_gemm_epilogues(prec, M, N, K)


def _check_layer_gemm_epilogues(prec, M, N, K):
    rt = Runtime(prec)
    A, Ar = to_op(rt, rnd(M, K, seed=1))
    B, Br = to_op(rt, rnd(N, K, seed=2, scale=K ** -0.5))
    bias = rnd(N, seed=3).to(DEV)
    lin = (Ar.to(DEV).double() @ Br.to(DEV).double().t()).cpu()
    ref = lin + bias.cpu().double()
    t = tol(prec, ref) * 2
    ld = _ru(N)
    out = torch.full((M, N), float("nan"), device=DEV)
    rt.gemm(L.EPI_STORE_F32, A, B, M, N, K, out, N, bias=bias)
    assert (out.cpu().double() - ref).abs().max().item() <= t
    oT = torch.zeros((M, ld), dtype=rt.op_dtype, device=DEV)
    rt.gemm(L.EPI_STORE_T, A, B, M, N, K, oT, ld, bias=bias)
    assert (oT[:, N:] == 0).all()
    assert (oT[:, :N].float().cpu().double() - ref).abs().max().item() <= t + HALF_ULP[prec] * float(ref.abs().max())
    rt.gemm(L.EPI_RELU_T, A, B, M, N, K, oT, ld, bias=bias)
    assert (oT[:, :N].float().cpu().double() - ref.clamp(min=0)).abs().max().item() <= t + HALF_ULP[prec] * float(ref.abs().max())
    res = rnd(M, N, seed=4).to(DEV)
    rt.gemm(L.EPI_ADD_F32, A, B, M, N, K, out, N, res=res, ldres=N)
    assert (out.cpu().double() - (lin + res.cpu().double())).abs().max().item() <= t
    sc = torch.tensor([0.25], device=DEV)
    rt.gemm(L.EPI_ADD_F32, A, B, M, N, K, out, N, res=res, ldres=N, acc_scale=L.ptr(sc))
    assert (out.cpu().double() - (0.25 * lin + res.cpu().double())).abs().max().item() <= t
    if N % 4 == 0:
        p, seed, site = 0.3, 1234567, 42
        mask = torch.empty((M, N), dtype=torch.uint8, device=DEV)
        L.call("timhip_dropout_mask", seed, site, p, M, N, L.ptr(mask), st())
        rt.gemm(L.EPI_DROP_RES_F32, A, B, M, N, K, out, N, bias=bias, res=res, ldres=N, p_drop=p, seed=seed, site=site)
        want = res.cpu().double() + ref * mask.cpu().double() / (1 - p)
        assert (out.cpu().double() - want).abs().max().item() <= t * 2
        aux = rnd(M, ld, seed=9).to(DEV).to(rt.op_dtype)
        rt.gemm(L.EPI_MULAUX_T, A, B, M, N, K, oT, ld, aux=aux, ldaux=ld)
        want = lin * aux[:, :N].float().cpu().double()
        assert (oT[:, :N].float().cpu().double() - want).abs().max().item() <= t * 4 + HALF_ULP[prec] * float(want.abs().max())
        # GELU + dropout with the keep-bits drawn ahead of time, two outputs (h = mask * gelu(u), g = mask * gelu'(u))
        ldm = (N + 7) // 8
        padded = torch.zeros((M, ldm * 8), dtype=torch.uint8, device=DEV)
        padded[:, :N] = mask
        bits = (padded.view(M, ldm, 8).to(torch.int32) << torch.arange(8, device=DEV, dtype=torch.int32)).sum(-1).to(torch.uint8)
        h = torch.zeros((M, ld), dtype=rt.op_dtype, device=DEV)
        g2 = torch.zeros((M, ld), dtype=rt.op_dtype, device=DEV)
        rt.gemm(L.EPI_GELU_DROP_G2, A, B, M, N, K, h, ld, out1=g2, ld1=ld, bias=bias, p_drop=p, seed=seed, site=site, mask=bits,
                ldmask=ldm)
        mk = mask.cpu().double() / (1 - p)
        u = ref.clone().requires_grad_(True)
        O._gelu(u).sum().backward()
        assert (h[:, :N].float().cpu().double() - O._gelu(ref) * mk).abs().max().item() <= t * 4 + HALF_ULP[prec] * float(ref.abs().max()) * 2
        assert (g2[:, :N].float().cpu().double() - u.grad * mk).abs().max().item() <= t * 4 + HALF_ULP[prec] * 4
    torch.cuda.synchronize()


@pytest.mark.parametrize("prec", H16)
def test_split_operand_gemm(prec):
    """timhip_split3_many: [hi | lo | hi] activations x [hi | hi | lo] weights through the plain 16-bit GEMM over K' = 3 K gives
    the fp32 product to ~2^-20 (what the fp16 model's time MLP and heads rely on); weights of the size the model has
    (|w| < 1/32: their lo parts are fp16 SUBNORMALS - the matrix pipe must honour them), ragged K"""
    rt = Runtime(prec)
    M, N, K = 960, 300, 1000
    x = rnd(M, K, seed=1) * 1.5
    w = (torch.rand(N, K, generator=torch.Generator().manual_seed(2)) * 2 - 1) / 32
    xd, wd = x.to(DEV), w.to(DEV)
    Kp = _ru(K)
    A3 = torch.empty((M, 3 * Kp), dtype=rt.op_dtype, device=DEV)
    B3 = torch.empty((N, 3 * Kp), dtype=rt.op_dtype, device=DEV)
    rt.split3([(xd, M, K, K, A3)], mode=0)
    rt.split3([(wd, N, K, K, B3)], mode=1)
    out = torch.empty((M, N), device=DEV)
    rt.gemm(L.EPI_STORE_F32, A3, B3, M, N, 3 * Kp, out, N)
    torch.cuda.synchronize()
    ref = x.double() @ w.double().t()
    plain = x.to(rt.op_dtype).double() @ w.to(rt.op_dtype).double().t()
    err = (out.cpu().double() - ref).abs().max().item()
    err_plain = (plain - ref).abs().max().item()
    lim = {"fp16": 3e-6, "bf16": 2e-4}[prec] * float(ref.abs().max())
    assert err <= lim, (err, err_plain)
    assert err_plain > 20 * err
    # relu variant and the layout of the blocks
    rt.split3([(xd, M, K, K, A3)], mode=0, relu=True)
    torch.cuda.synchronize()
    hi = A3[:, :K].float().cpu()
    assert torch.equal(hi, x.clamp(min=0).to(rt.op_dtype).float())
    assert torch.equal(A3[:, 2 * Kp:2 * Kp + K].float().cpu(), hi)
    assert (A3[:, K:Kp] == 0).all() and (A3[:, Kp + K:2 * Kp] == 0).all()
    # unaligned source (row stride and width not multiples of 4): the element-wise path of the kernel, two items in one launch
    wide = (rnd(37, 1010, seed=5) * 0.7).to(DEV)
    src = wide[:, 3:1005]                       # 1002 columns, row stride 1010, base address 12 bytes off
    A4 = torch.empty((37, 3 * Kp), dtype=rt.op_dtype, device=DEV)
    A5 = torch.empty((M, 3 * Kp), dtype=rt.op_dtype, device=DEV)
    rt.split3([(src, 37, 1002, 1010, A4), (xd, M, K, K, A5)], mode=0)
    torch.cuda.synchronize()
    hi4 = src.to(rt.op_dtype)
    assert torch.equal(A4[:, :1002], hi4) and torch.equal(A4[:, 2 * Kp:2 * Kp + 1002], hi4)
    assert torch.equal(A4[:, Kp:Kp + 1002], (src - hi4.float()).to(rt.op_dtype))
    assert (A4[:, 1002:Kp] == 0).all()
    assert torch.equal(A5[:, :K].float().cpu(), x.to(rt.op_dtype).float())


@pytest.mark.parametrize("prec", H16)
@pytest.mark.parametrize("M,N,K", [(9920, 1024, 1024), (333, 256, 128), (64, 72, 64)])
def test_gemm_weight_split_with_wrapped_a_operand(prec, M, N, K):
    """TimEpi.a_wrap_k (round 3): C = [A | A] [w_hi | w_lo]^T - the A operand read twice along a contraction of 2K, the weight as
    the first two column blocks of its split copy.  The product must carry the fp32 weight to ~20 bits (against 2^-9 / 2^-12 for
    the plain 16-bit weight), on the ping-pong kernel (first shape) and the two-blocks-per-CU kernel; with the fused residual
    epilogue the encoder layer's out-projection uses"""
    rt = Runtime(prec)
    A, Ar = to_op(rt, rnd(M, K, seed=1))
    w = (rnd(N, K, seed=2, scale=K ** -0.5)).to(DEV)
    W3 = torch.empty((N, 3 * _ru(K)), dtype=rt.op_dtype, device=DEV)
    rt.split3([(w, N, K, K, W3)], mode=0)
    ref = Ar.to(DEV).double() @ w.double().t()
    out = torch.full((M, N), float("nan"), device=DEV)
    rt.gemm(L.EPI_STORE_F32, A, W3, M, N, 2 * K, out, N, rep=2, a_wrap_k=K)
    torch.cuda.synchronize()
    err = (out.double() - ref).abs().max().item()
    plain = (Ar.to(DEV).double() @ w.to(rt.op_dtype).double().t() - ref).abs().max().item()
    lim = {"fp16": 3e-6, "bf16": 3e-5}[prec] * max(1.0, ref.abs().max().item())
    assert err <= lim, (err, plain)
    assert err <= 0.1 * plain
    res = rnd(M, N, seed=4).to(DEV)
    rt.gemm(L.EPI_DROP_RES_F32, A, W3, M, N, 2 * K, out, N, res=res, ldres=N, rep=2, a_wrap_k=K)
    torch.cuda.synchronize()
    assert (out.double() - (ref + res.double())).abs().max().item() <= lim * 2


def test_dx_init_writes_the_whole_stream():
    """timhip_dx_init (round 4): the gradient stream entering the encoder stack in one pass - feature rows <- the `feats`
    cotangent (zeros without one), the rows of up to six DISJOINT token ranges <- their heads' rows, every other row zero;
    overlapping ranges and ranges that reach into the feature rows are refused (the caller adds those instead)."""
    B, S, F, E = 3, 23, 9, 64
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(B, F, E, generator=g).to(DEV)
    ranges = [(11, 4), (18, 5), (9, 2)]           # [9,11) [11,15) [18,23): rows 15..17 belong to nobody
    rows = [torch.randn(B * n, E, generator=g).to(DEV) for _, n in ranges]
    want = torch.zeros(B, S, E, device=DEV)
    want[:, :F] = feats
    for (s0, n), r in zip(ranges, rows):
        want[:, s0:s0 + n] = r.view(B, n, E)
    dx = torch.full((B * S, E), float("nan"), device=DEV)
    ia = lambda v: (C.c_int * len(v))(*v)
    pa = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    L.call("timhip_dx_init", B, S, F, E, L.ptr(feats), 3, ia([r[0] for r in ranges]), ia([r[1] for r in ranges]), pa(rows), L.ptr(dx), st())
    torch.cuda.synchronize()
    assert torch.equal(dx.view(B, S, E), want)
    dx.fill_(float("nan"))
    L.call("timhip_dx_init", B, S, F, E, None, 0, None, None, None, L.ptr(dx), st())      # nothing to copy: all zeros
    torch.cuda.synchronize()
    assert bool((dx == 0).all())
    lib = L.load()
    bad = lib.timhip_dx_init(B, S, F, E, L.ptr(feats), 2, ia([11, 13]), ia([4, 4]), pa(rows[:2]), L.ptr(dx), st())   # overlap
    assert bad != 0
    bad = lib.timhip_dx_init(B, S, F, E, L.ptr(feats), 1, ia([F - 1]), ia([4]), pa(rows[:1]), L.ptr(dx), st())       # into the feature rows
    assert bad != 0


def test_dx_init_adds_up_slabs():
    """timhip_dx_init_slabs (round 5): a token range whose head rows arrive as several [B n, E] slabs - the column chunks of a
    long-contraction input-gradient product - is the SUM of its slabs; ranges with one slab as before."""
    B, S, F, E = 3, 23, 9, 64
    g = torch.Generator().manual_seed(6)
    feats = torch.randn(B, F, E, generator=g).to(DEV)
    ranges = [(11, 4, 3), (18, 5, 1), (9, 2, 2)]
    rows = [torch.randn(ns, B * n, E, generator=g).to(DEV) for _, n, ns in ranges]
    want = torch.zeros(B, S, E, device=DEV)
    want[:, :F] = feats
    for (s0, n, ns), r in zip(ranges, rows):
        acc = r[0].clone()
        for z in range(1, ns):
            acc += r[z]                                # the kernel's order of additions
        want[:, s0:s0 + n] = acc.view(B, n, E)
    dx = torch.full((B * S, E), float("nan"), device=DEV)
    ia = lambda v: (C.c_int * len(v))(*v)
    pa = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    L.call("timhip_dx_init_slabs", B, S, F, E, L.ptr(feats), 3, ia([r[0] for r in ranges]), ia([r[1] for r in ranges]), pa(rows),
           ia([r[2] for r in ranges]), L.ptr(dx), st())
    torch.cuda.synchronize()
    assert torch.equal(dx.view(B, S, E), want)
    assert L.load().timhip_dx_init_slabs(B, S, F, E, L.ptr(feats), 1, ia([11]), ia([4]), pa(rows[:1]), ia([0]), L.ptr(dx), st()) != 0


# ---- round 5: the fused / paired entry points against the launches they replace (bit for bit) ----------------------------------
@pytest.mark.parametrize("prec", ["fp16", "bf16", "fp32"])
def test_cast_rows_pair_is_two_cast_rows(prec):
    """timhip_cast_rows_pair: two feature matrices of different widths, one launch - the same operand rows and the same dropout
    masks (per site) as two timhip_cast_rows calls"""
    rt = Runtime(prec)
    R, cols, sites, seed, p = 96, [24, 40], [L.SITE_FEAT_V, L.SITE_FEAT_A], 0x1234567, 0.5
    src = [rnd(R, c, seed=3 + i).to(DEV) for i, c in enumerate(cols)]
    want = [torch.empty((R, _ru(c)), dtype=rt.op_dtype, device=DEV) for c in cols]
    got = [torch.full((R, _ru(c)), 7.0, dtype=rt.op_dtype, device=DEV) for c in cols]
    for s_, c, w, site in zip(src, cols, want, sites):
        L.call("timhip_cast_rows", rt.prec, L.ptr(s_), R, c, c, L.ptr(w), w.shape[1], p, seed, site, None, st())
    pa = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    ia = lambda v: (C.c_int * len(v))(*v)
    L.call("timhip_cast_rows_pair", rt.prec, pa(src), ia(cols), pa(got), ia([g.shape[1] for g in got]), R, p, seed,
           (C.c_uint32 * 2)(*sites), st())
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        assert torch.equal(w, g)
    assert not torch.equal(got[0][:, :24], got[1][:, :24])


@pytest.mark.parametrize("R", [160, 8000])
@pytest.mark.parametrize("prec", ["fp16", "fp32"])
def test_layernorm_pair_is_two_layernorms(prec, R):
    """timhip_layernorm_fwd2 / _bwd2: two LayerNorms (GELU in front, as the modality embedders) over stacked rows with a second
    parameter set from the split row on - outputs, statistics, operand-dtype gradients and both parameter-gradient pairs equal two
    separate launches (the parameter gradients up to the order of their atomics).  R = 8000 (B = 160 windows of 50 feature tokens,
    16000 stacked rows): past 12288 rows the backward's balanced-round choice is 24 rows per block, which does not divide 8000 -
    the launch falls back to 16-row blocks instead of refusing (round-5 advisor finding)"""
    rt = Runtime(prec)
    d = 512
    u = rnd(2 * R, d, seed=1).to(DEV)
    g = rnd(2 * R, d, seed=2).to(DEV)
    ws = [(1.0 + 0.1 * rnd(d, seed=3 + i)).to(DEV) for i in range(2)]
    bs = [(0.1 * rnd(d, seed=5 + i)).to(DEV) for i in range(2)]
    e_w, st_w = torch.empty(2 * R, d, device=DEV), torch.empty(2 * R, 2, device=DEV)
    e_g, st_g = torch.empty_like(e_w), torch.empty_like(st_w)
    for i in range(2):
        sl = slice(i * R, (i + 1) * R)
        rt.ln_fwd(u[sl], R, d, 2, ws[i], bs[i], xf=e_w[sl], ldx=d, stats=st_w[sl])
    L.call("timhip_layernorm_fwd2", rt.prec, L.ptr(u), 2 * R, d, d, 2, L.ptr(ws[0]), L.ptr(bs[0]), R, L.ptr(ws[1]), L.ptr(bs[1]),
           L.ptr(e_g), d, None, 0, L.ptr(st_g), st())
    torch.cuda.synchronize()
    assert torch.equal(e_w, e_g) and torch.equal(st_w, st_g)
    dy_w = torch.zeros((2 * R, d), dtype=rt.op_dtype, device=DEV)
    dy_g = torch.zeros_like(dy_w)
    dgw = [torch.zeros(d, device=DEV) for _ in range(4)]
    dgg = [torch.zeros(d, device=DEV) for _ in range(4)]
    for i in range(2):
        sl = slice(i * R, (i + 1) * R)
        rt.ln_bwd(g[sl], u[sl], st_w[sl], R, d, 2, ws[i], dyt=dy_w[sl], dgamma=dgw[2 * i], dbeta=dgw[2 * i + 1])
    L.call("timhip_layernorm_bwd2", rt.prec, L.ptr(g), d, L.ptr(u), d, L.ptr(st_g), 2 * R, d, 2, L.ptr(ws[0]), R, L.ptr(ws[1]), None, 0,
           L.ptr(dy_g), d, L.ptr(dgg[0]), L.ptr(dgg[1]), L.ptr(dgg[2]), L.ptr(dgg[3]), None, st())
    torch.cuda.synchronize()
    assert torch.equal(dy_w, dy_g)
    for a, b in zip(dgw, dgg):
        assert (a - b).abs().max().item() <= (1e-5 if R < 1000 else 1e-4) * a.abs().max().item()
    assert L.load().timhip_layernorm_bwd2(rt.prec, L.ptr(g), d, L.ptr(u), d, L.ptr(st_g), 2 * R, d, 2, L.ptr(ws[0]), R + 2, L.ptr(ws[1]),
                                          None, 0, L.ptr(dy_g), d, L.ptr(dgg[0]), L.ptr(dgg[1]), L.ptr(dgg[2]), L.ptr(dgg[3]), None,
                                          st()) != 0      # the halves must meet at a multiple of 4 rows (one row per wave and pass)


@pytest.mark.parametrize("prec", H16)
@pytest.mark.parametrize("d", [32, 512])
def test_split_operands_written_by_their_producers(prec, d):
    """round 5: time-MLP layer 1 (timhip_time_l1_fwd_split3), the relu + split epilogue (TIMHIP_EPI_RELU_SPLIT3_T) and the heads'
    row gather (timhip_gather_split3_ranges) write [hi | lo | hi] blocks themselves - bit for bit what timhip_split3_many (mode 0)
    made of the fp32 values in a second launch"""
    rt = Runtime(prec)
    R, ldd = 200, _ru(d)
    t2 = torch.rand(R, 2, generator=torch.Generator().manual_seed(1)).to(DEV)
    w0, b0 = rnd(d, 2, seed=2).to(DEV), rnd(d, seed=3).to(DEV)
    f1 = torch.empty(R, d, device=DEV)
    L.call("timhip_time_l1_fwd", L.PREC_FP32, L.ptr(t2), R, d, L.ptr(w0), L.ptr(b0), L.ptr(f1), d, st())
    want = torch.zeros((R, 3 * ldd), dtype=rt.op_dtype, device=DEV)
    rt.split3([(f1, R, d, d, want)], mode=0)
    got = torch.full((R, 3 * ldd), 5.0, dtype=rt.op_dtype, device=DEV)
    L.call("timhip_time_l1_fwd_split3", rt.prec, L.ptr(t2), R, d, L.ptr(w0), L.ptr(b0), L.ptr(got), ldd, st())
    torch.cuda.synchronize()
    assert torch.equal(want, got)
    # GEMM epilogue: relu(A W^T + b) as split blocks
    A, _ = to_op(rt, rnd(R, 64, seed=4))
    W, _ = to_op(rt, rnd(d, 64, seed=5, scale=0.2))
    bias = rnd(d, seed=6).to(DEV)
    f2 = torch.empty(R, d, device=DEV)
    rt.gemm(L.EPI_STORE_F32, A, W, R, d, 64, f2, d, bias=bias)
    want2 = torch.zeros((R, 3 * ldd), dtype=rt.op_dtype, device=DEV)
    rt.split3([(f2, R, d, d, want2)], mode=0, relu=True)
    got2 = torch.zeros((R, 3 * ldd), dtype=rt.op_dtype, device=DEV)
    rt.gemm(L.EPI_RELU_SPLIT3_T, A, W, R, d, 64, got2, 3 * ldd, ld1=ldd, bias=bias)
    torch.cuda.synchronize()
    assert torch.equal(want2, got2)
    if d % 64 == 0:   # heads: rows gathered from the fp32 stream and split in one pass
        B, S, E = 3, 11, d
        x = rnd(B * S, E, seed=7).to(DEV)
        ranges = [(4, 3), (8, 2)]
        ia = lambda v: (C.c_int * len(v))(*v)
        pa = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        rows = [torch.empty(B * n, E, device=DEV) for _, n in ranges]
        L.call("timhip_gather_ranges", L.PREC_FP32, L.ptr(x), B, S, E, 2, ia([r[0] for r in ranges]), ia([r[1] for r in ranges]), pa(rows), st())
        want3 = [torch.empty((B * n, 3 * E), dtype=rt.op_dtype, device=DEV) for _, n in ranges]
        rt.split3([(r_, B * n, E, E, w_) for r_, (_, n), w_ in zip(rows, ranges, want3)], mode=0)
        got3 = [torch.empty((B * n, 3 * E), dtype=rt.op_dtype, device=DEV) for _, n in ranges]
        L.call("timhip_gather_split3_ranges", rt.prec, L.ptr(x), B, S, E, 2, ia([r[0] for r in ranges]), ia([r[1] for r in ranges]), pa(got3), st())
        torch.cuda.synchronize()
        for a, b in zip(want3, got3):
            assert torch.equal(a, b)
