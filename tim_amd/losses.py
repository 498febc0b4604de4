"""Loss tail of the TIM training step on the HIP kernels.

Mirrors the reference's host functions by name and argument meaning:
  * `dense_relative_localization_loss(x, model, m)`, `dense_relative_localization_loss_crossmodal(x1, x2, model, m)`,
    `position_sampling`, `collect_samples`          <- time_interval_machine/models/helpers/losses/drloc.py:4-41
  * `mixup_criterion(criterion, pred_a, pred_b, y_a, y_b, lam, weights=None)`  <- utils/mixup.py:24-39
  * `CrossEntropyLoss(label_smoothing=0.2, ignore_index=-1)`                   <- the criterion of scripts/train.py:46-49
plus `mixup_cross_entropy(logits, target_a, target_b, lam)`, the one-pass form of what train.py:218-316 spells as
"filter the valid rows twice, run the criterion twice, blend": one read of the logits for the loss, one write of the
gradient.  Everything here runs on libtimhip (tim_amd/csrc/losses.hip and the GEMM kernels); there is no CPU path.
"""
import ctypes as C

import torch

from . import _lib as L
from ._lib import call, ptr
from .functional import _f32c, _require_gpu, _ru, _stream


# ==================================================================================================
# label-smoothed cross entropy (+ mixup)
# ==================================================================================================
class _MixupCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target_a, target_b, lam, smoothing):
        _require_gpu(logits, "cross entropy")
        x = _f32c(logits)
        if x.dim() != 2:
            raise ValueError("logits must be [rows, classes]")
        rows, Cn = x.shape
        dev = x.device
        ta = target_a.to(device=dev, dtype=torch.int64).contiguous().view(-1)
        tb = None if target_b is None else target_b.to(device=dev, dtype=torch.int64).contiguous().view(-1)
        if ta.numel() != rows or (tb is not None and tb.numel() != rows):
            raise ValueError("one target per logits row")
        loss = torch.empty((), dtype=torch.float32, device=dev)
        if rows == 0:
            ctx.empty = True
            ctx.shape = tuple(x.shape)
            return loss.zero_()
        stats = torch.empty((rows, 4), dtype=torch.float32, device=dev)
        accum = torch.empty(4, dtype=torch.float32, device=dev)
        call("timhip_ce_mixup_fwd", ptr(x), rows, Cn, x.stride(0), ptr(ta), ptr(tb), float(lam), float(smoothing),
             ptr(stats), ptr(accum), ptr(loss), _stream())
        ctx.empty = False
        ctx.lam, ctx.smoothing = float(lam), float(smoothing)
        ctx.save_for_backward(x, ta, tb if tb is not None else ta.new_empty(0), stats, accum)
        ctx.has_b = tb is not None
        return loss

    @staticmethod
    def backward(ctx, g):
        if ctx.empty:
            return torch.zeros(ctx.shape, dtype=torch.float32, device=g.device), None, None, None, None
        x, ta, tb, stats, accum = ctx.saved_tensors
        rows, Cn = x.shape
        gout = _f32c(g).reshape(1)
        dx = torch.empty_like(x)
        call("timhip_ce_mixup_bwd", ptr(x), rows, Cn, x.stride(0), ptr(ta), ptr(tb) if ctx.has_b else None, ctx.lam,
             ctx.smoothing, ptr(stats), ptr(accum), ptr(gout), ptr(dx), dx.stride(0), _stream())
        return dx, None, None, None, None


def mixup_cross_entropy(logits, target_a, target_b, lam, label_smoothing=0.2):
    """lam * CE(logits[target_a != -1], target_a[...]).mean() + (1-lam) * CE(logits[target_b != -1], target_b[...]).mean()
    with label smoothing - the value train.py:243-258 computes per head - in one pass over `logits`."""
    return _MixupCEFn.apply(logits, target_a, target_b, lam, label_smoothing)


class CrossEntropyLoss(torch.nn.Module):
    """`torch.nn.CrossEntropyLoss(label_smoothing=..., ignore_index=-1)` (mean over the non-ignored rows) on the HIP kernel."""

    def __init__(self, label_smoothing=0.0, ignore_index=-1):
        super().__init__()
        if ignore_index != -1:
            raise ValueError("the TIM training loop uses ignore_index=-1 (train.py:48)")
        self.label_smoothing = float(label_smoothing)
        self.ignore_index = ignore_index

    def forward(self, logits, target):
        return _MixupCEFn.apply(logits, target, None, 1.0, self.label_smoothing)


def mixup_criterion(criterion, pred_a, pred_b, y_a, y_b, lam, weights=None):
    """utils/mixup.py:24-39, same arguments.  (`weights` scales the two terms exactly as the reference does.)"""
    loss_a = criterion(pred_a, y_a)
    loss_b = criterion(pred_b, y_b)
    if weights is not None:
        wa, wb = (weights[0], weights[1]) if isinstance(weights, (list, tuple)) else (weights, weights)
        loss_a, loss_b = loss_a * wa, loss_b * wb
    return lam * loss_a.mean() + (1 - lam) * loss_b.mean()


# ==================================================================================================
# DRLoc
# ==================================================================================================
def _mlp_fwd(rt, xT, R, w0, b0, w2, b2, w4, b4):
    """Linear(4d,d) ReLU Linear(d,d) ReLU Linear(d,1) (tim.py:129-135) on an operand-dtype input [R, ru(4d)]"""
    dev = xT.device
    d, K0 = w0.shape
    ldd = _ru(d)
    h1 = torch.zeros((R, ldd), dtype=rt.op_dtype, device=dev)
    rt.gemm(L.EPI_RELU_T, xT, rt.weight(w0), R, d, K0, h1, ldd, bias=_f32c(b0))
    h2 = torch.zeros((R, ldd), dtype=rt.op_dtype, device=dev)
    rt.gemm(L.EPI_RELU_T, h1, rt.weight(w2), R, d, d, h2, ldd, bias=_f32c(b2))
    y = torch.empty((R, 1), dtype=torch.float32, device=dev)
    rt.gemm(L.EPI_STORE_F32, h2, rt.weight(w4), R, 1, d, y, 1, bias=_f32c(b4))
    return h1, h2, y


def _mlp_bwd(rt, gy, xT, h1, h2, w0, w2, w4, need_dx):
    dev = xT.device
    d, K0 = w0.shape
    R = xT.shape[0]
    st = _stream()
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    dw0, db0, dw2, db2, dw4, db4 = z(d, K0), z(d), z(d, d), z(d), z(1, d), z(1)
    g = _f32c(gy).reshape(R, 1)
    # fp16: the cotangent of a mean-reduced L1 loss is ~1e-4 / 1e-5 and the chain multiplies it with weights of ~0.04 - fp16
    # subnormals.  As in the encoder, the 16-bit gradient operands carry a power-of-two factor chosen on the device from the
    # incoming cotangent; parameter gradients and the fp32 input gradient are true-scale (timhip_grad_scale, include/timhip.h)
    gs = rt.grad_scale([g], dev)
    gs_in, gs_out = (ptr(gs), ptr(gs) + 4) if gs is not None else (None, None)
    gT = torch.empty((R, 64), dtype=rt.op_dtype, device=dev)
    call("timhip_cast_rows", rt.prec, ptr(g), R, 1, 1, ptr(gT), 64, 0.0, 0, 0, gs_in, st)
    dh2 = torch.zeros_like(h2)
    rt.gemm(L.EPI_DRELU_T, gT, rt.weight(w4, True), R, d, 1, dh2, dh2.shape[1], aux=h2, ldaux=h2.shape[1])
    dh1 = torch.zeros_like(h1)
    rt.gemm(L.EPI_DRELU_T, dh2, rt.weight(w2, True), R, d, d, dh1, dh1.shape[1], aux=h1, ldaux=h1.shape[1])
    rt.wgrad_many([(gT, 1, h2, d, R, dw4, db4), (dh2, d, h1, d, R, dw2, db2), (dh1, d, xT, K0, R, dw0, db0)], out_scale=gs_out)
    dpts = None
    if need_dx:
        dpts = torch.empty((R, K0), dtype=torch.float32, device=dev)
        rt.gemm(L.EPI_ADD_F32, dh1, rt.weight(w0, True), R, K0, d, dpts, K0, acc_scale=gs_out)   # (no residual: out0 = acc / S)
    return dpts, (dw0, db0, dw2, db2, dw4, db4)


class DrlocMlpFn(torch.autograd.Function):
    """`model(x, "drloc_mlp")` on an fp32 input [.., 4d]"""

    @staticmethod
    def forward(ctx, rt, x, w0, b0, w2, b2, w4, b4):
        _require_gpu(x, "drloc_mlp")
        K0 = w0.shape[1]
        x2 = _f32c(x).reshape(-1, K0)
        R = x2.shape[0]
        xT = torch.empty((R, _ru(K0)), dtype=rt.op_dtype, device=x.device)
        if _ru(K0) != K0:
            xT.zero_()
        call("timhip_cast_rows", rt.prec, ptr(x2), R, K0, K0, ptr(xT), xT.shape[1], 0.0, 0, 0, None, _stream())
        h1, h2, y = _mlp_fwd(rt, xT, R, w0, b0, w2, b2, w4, b4)
        ctx.rt, ctx.xshape = rt, tuple(x.shape)
        ctx.save_for_backward(xT, h1, h2, w0, w2, w4)
        return y

    @staticmethod
    def backward(ctx, gy):
        xT, h1, h2, w0, w2, w4 = ctx.saved_tensors
        dpts, gw = _mlp_bwd(ctx.rt, gy, xT, h1, h2, w0, w2, w4, ctx.needs_input_grad[1])
        return (None, None if dpts is None else dpts.view(ctx.xshape)) + gw


class DrlocFn(torch.autograd.Function):
    """collect_samples of both points + cat + drloc_mlp (drloc.py:24-26 / 37-39) with the gather writing the first
    GEMM's operand directly and the backward scattering the operand gradient into d x1 / d x2."""

    @staticmethod
    def forward(ctx, rt, x1, x2, pos1, pos2, m, w0, b0, w2, b2, w4, b4):
        _require_gpu(x1, "drloc")
        n, l, D = x1.shape
        K0 = 2 * D
        if w0.shape[1] != K0:
            raise ValueError("drloc_mlp.0 expects %d inputs, the sampled pairs have %d" % (w0.shape[1], K0))
        same = x2 is x1 or (x1.data_ptr() == x2.data_ptr() and x1.shape == x2.shape and x1.stride() == x2.stride())
        ok = lambda t: t.dtype == torch.float32 and t.stride(2) == 1 and t.stride(1) % 4 == 0 and t.stride(0) % 4 == 0 \
            and t.data_ptr() % 16 == 0
        a = x1.detach() if ok(x1) else x1.detach().float().contiguous()
        b = a if same else (x2.detach() if ok(x2) else x2.detach().float().contiguous())
        if a.stride() != b.stride():
            a, b = a.contiguous(), b.contiguous()
        pts = torch.empty((n * m, _ru(K0)), dtype=rt.op_dtype, device=x1.device)
        if _ru(K0) != K0:
            pts.zero_()
        call("timhip_drloc_gather", rt.prec, ptr(a), ptr(b), a.stride(0), a.stride(1), n, l, D, ptr(pos1), ptr(pos2), m,
             ptr(pts), pts.shape[1], _stream())
        h1, h2, y = _mlp_fwd(rt, pts, n * m, w0, b0, w2, b2, w4, b4)
        ctx.rt, ctx.same, ctx.m, ctx.shape = rt, same, m, (n, l, D)
        ctx.save_for_backward(pts, h1, h2, w0, w2, w4, pos1, pos2)
        return y

    @staticmethod
    def backward(ctx, gy):
        pts, h1, h2, w0, w2, w4, pos1, pos2 = ctx.saved_tensors
        need = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dpts, gw = _mlp_bwd(ctx.rt, gy, pts, h1, h2, w0, w2, w4, need)
        dx1 = dx2 = None
        if need:
            n, l, D = ctx.shape
            dx1 = torch.zeros((n, l, D), dtype=torch.float32, device=pts.device)
            dx2 = dx1 if ctx.same else torch.zeros_like(dx1)
            call("timhip_drloc_scatter_add", ptr(dpts), dpts.shape[1], ptr(dx1), ptr(dx2), l * D, D, n, l, D, ptr(pos1),
                 ptr(pos2), ctx.m, _stream())
            if ctx.same:
                dx2 = None
        return (None, dx1, dx2, None, None, None) + gw


def drloc_mlp_forward(model, inputs):
    """`model(inputs, "drloc_mlp")` (tim.py:190-191): [n, m, 4d] -> [n, m]"""
    mlp = model.drloc_mlp
    y = DrlocMlpFn.apply(model.rt, inputs, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, mlp[4].weight,
                         mlp[4].bias)
    return y.view(*inputs.shape[:-1], 1).squeeze(2)


def position_sampling(k, m, n):
    """drloc.py:4-7 (host RNG, same draw order)"""
    pos_1 = torch.randint(k, size=(n, m))
    pos_2 = torch.randint(k, size=(n, m))
    return pos_1, pos_2


def collect_samples(x, pos, n):
    """drloc.py:10-14 (kept for API parity; the losses below gather with timhip_drloc_gather instead)"""
    _, l, D = x.size()
    x = x.permute(2, 0, 1).reshape(D, -1)
    pos = ((torch.arange(n).long().to(pos.device) * l).view(n, 1) + pos).view(-1)
    return (x[:, pos]).view(D, n, -1).permute(1, 0, 2)


def _drloc(x1, x2, model, m, positions):
    core = getattr(model, "module", model)
    n, l, D = x1.size()
    pos_1, pos_2 = position_sampling(l, m, n) if positions is None else positions
    deltax = torch.abs((pos_1 - pos_2).float()).to(x1.device)
    deltax /= l
    p1 = pos_1.to(device=x1.device, dtype=torch.int64).contiguous()
    p2 = pos_2.to(device=x1.device, dtype=torch.int64).contiguous()
    mlp = core.drloc_mlp
    y = DrlocFn.apply(core.rt, x1, x2, p1, p2, m, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, mlp[4].weight,
                      mlp[4].bias)
    return torch.nn.functional.l1_loss(deltax, y.view(n, m))


def dense_relative_localization_loss(x, model, m, positions=None):
    """drloc.py:17-27.  `positions=(pos_1, pos_2)` pins the sampled pairs (tests); default: drawn as the reference does."""
    return _drloc(x, x, model, m, positions)


def dense_relative_localization_loss_crossmodal(x1, x2, model, m, positions=None):
    """drloc.py:30-41 (train.py:331-336 passes output[1][:, :num_feats] and output[1][:, num_feats:])"""
    assert x1.size() == x2.size()
    return _drloc(x1, x2, model, m, positions)


# ==================================================================================================
# detection losses
# ==================================================================================================
class _FocalSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, row_weights, row_valid, alpha, gamma):
        _require_gpu(logits, "sigmoid focal loss")
        x = _f32c(logits)
        t = _f32c(targets).to(x.device)
        if x.shape != t.shape:
            raise ValueError("Mismatch in input and target shape: %s != %s" % (tuple(x.shape), tuple(t.shape)))
        Cn = x.shape[-1] if x.dim() > 1 else 1
        rows = x.numel() // max(Cn, 1)
        w = None if row_weights is None else _f32c(row_weights).to(x.device).reshape(-1)
        v = None if row_valid is None else row_valid.to(device=x.device, dtype=torch.uint8).contiguous().reshape(-1)
        if (w is not None and w.numel() != rows) or (v is not None and v.numel() != rows):
            raise ValueError("one weight / valid flag per row")
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        call("timhip_focal_loss_fwd", ptr(x), ptr(t), rows, Cn, ptr(w), ptr(v), float(alpha), float(gamma), ptr(loss),
             None, _stream())
        ctx.alpha, ctx.gamma, ctx.rows, ctx.C = float(alpha), float(gamma), rows, Cn
        ctx.save_for_backward(x, t, w if w is not None else x.new_empty(0), v if v is not None else x.new_empty(0, dtype=torch.uint8))
        ctx.has = (w is not None, v is not None)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, t, w, v = ctx.saved_tensors
        dx = torch.empty_like(x)
        gout = _f32c(g).reshape(1)
        call("timhip_focal_loss_bwd", ptr(x), ptr(t), ctx.rows, ctx.C, ptr(w) if ctx.has[0] else None,
             ptr(v) if ctx.has[1] else None, ctx.alpha, ctx.gamma, ptr(gout), ptr(dx), _stream())
        return dx, None, None, None, None, None


def focal_loss_sum(logits, targets, row_weights=None, row_valid=None, alpha=0.25, gamma=2.0):
    """sum_{r: row_valid[r]} row_weights[r] * sum_c focal(logits[r,c], targets[r,c]): the value det train.py:235-262 gets from
    `get_loss(sigmoid_focal_loss, preds[valid], targets[valid], weights=ious, reduction="sum")`, without the filtering."""
    return _FocalSumFn.apply(logits, targets, row_weights, row_valid, alpha, gamma)


def sigmoid_focal_loss(inputs, targets, alpha: float = 0.25, gamma: float = 2.0, reduction: str = "none"):
    """detection models/helpers/losses/sigmoid.py:5-52, same arguments"""
    if reduction == "sum":
        return focal_loss_sum(inputs, targets, None, None, alpha, gamma)
    if reduction == "mean":
        return focal_loss_sum(inputs, targets, None, None, alpha, gamma) / max(inputs.numel(), 1)
    return _focal_elementwise(inputs, targets, None, alpha, gamma)


def _focal_elementwise(inputs, targets, weights, alpha, gamma):
    """reduction "none" (meters only: no gradient)"""
    _require_gpu(inputs, "sigmoid focal loss")
    x = _f32c(inputs)
    t = _f32c(targets).to(x.device)
    Cn = x.shape[-1] if x.dim() > 1 else 1
    rows = x.numel() // max(Cn, 1)
    w = None if weights is None else _f32c(weights).to(x.device).reshape(-1)
    out = torch.empty_like(x)
    tot = torch.empty((), dtype=torch.float32, device=x.device)
    call("timhip_focal_loss_fwd", ptr(x), ptr(t), rows, Cn, ptr(w), None, float(alpha), float(gamma), ptr(tot), ptr(out),
         _stream())
    return out


class _DiouSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, row_valid, eps):
        _require_gpu(pred, "ctr_diou_loss_1d")
        p = _f32c(pred).reshape(-1, 2)
        t = _f32c(target).to(p.device).reshape(-1, 2)
        v = None if row_valid is None else row_valid.to(device=p.device, dtype=torch.uint8).contiguous().reshape(-1)
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        call("timhip_diou_1d", ptr(p), ptr(t), p.shape[0], ptr(v), float(eps), None, ptr(loss), None, _stream())
        ctx.eps, ctx.has_v = float(eps), v is not None
        ctx.save_for_backward(p, t, v if v is not None else p.new_empty(0, dtype=torch.uint8))
        ctx.shape = tuple(pred.shape)
        return loss

    @staticmethod
    def backward(ctx, g):
        p, t, v = ctx.saved_tensors
        dp = torch.empty_like(p)
        gout = _f32c(g).reshape(1)
        call("timhip_diou_1d", ptr(p), ptr(t), p.shape[0], ptr(v) if ctx.has_v else None, ctx.eps, ptr(gout), None, ptr(dp),
             _stream())
        return dp.view(ctx.shape), None, None, None


def diou_loss_sum(pred_offsets, target_offsets, row_valid=None, eps=1e-8):
    """sum over the valid rows of the 1-D centre-offset DIoU loss (det train.py:277-285 without the row filtering)"""
    return _DiouSumFn.apply(pred_offsets, target_offsets, row_valid, eps)


def ctr_diou_loss_1d(input_offsets, target_offsets, reduction: str = "none", eps: float = 1e-8):
    """detection models/helpers/losses/iou.py:4-65, same arguments ("none" is not needed by the training loop)"""
    if reduction == "sum":
        return diou_loss_sum(input_offsets, target_offsets, None, eps)
    if reduction == "mean":
        n = input_offsets.shape[0]
        s = diou_loss_sum(input_offsets, target_offsets, None, eps)
        return s / n if n > 0 else 0.0 * s
    raise NotImplementedError('ctr_diou_loss_1d: reduction "none" is not built (the training loop uses "sum")')


class _DetSideLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aux, iou, offsets, reg_pred, *logits):
        targets, normaliser, thr, alpha, gamma, eps, lam, mom = aux
        _require_gpu(reg_pred, "detection side loss")
        dev = reg_pred.device
        xs = [_f32c(x) for x in logits]
        ts = [_f32c(t).to(dev) for t in targets]
        rows = iou.numel()
        for x, t in zip(xs, ts):
            if x.shape != t.shape or x.dim() != 2 or x.shape[0] != rows:
                raise ValueError("logits / targets of one side must be [rows, C] with one row per query: %s vs %s, %d rows"
                                 % (tuple(x.shape), tuple(t.shape), rows))
        u = _f32c(iou).reshape(-1)
        off = _f32c(offsets).reshape(-1, 2)
        rp = _f32c(reg_pred).reshape(-1, 2)
        if off.shape[0] != rows or rp.shape[0] != rows:
            raise ValueError("one offset pair per query row")
        block = torch.empty(8, dtype=torch.float32, device=dev)
        Cs = (C.c_int * len(xs))(*[x.shape[1] for x in xs])
        pa = lambda ts_: (C.c_void_p * len(ts_))(*[ptr(t) for t in ts_])   # noqa: E731
        call("timhip_det_side_loss_fwd", pa(xs), pa(ts), Cs, len(xs), rows, ptr(u), ptr(off), ptr(rp), float(thr), float(alpha),
             float(gamma), float(eps), float(lam), float(mom), ptr(normaliser), ptr(block), _stream())
        ctx.consts = (float(thr), float(alpha), float(gamma), float(eps), float(lam), rows, tuple(reg_pred.shape))
        ctx.n = len(xs)
        ctx.save_for_backward(block, u, off, rp, *xs, *ts)
        return block[0]

    @staticmethod
    def backward(ctx, g):
        block, u, off, rp, *rest = ctx.saved_tensors
        xs, ts = rest[:ctx.n], rest[ctx.n:]
        thr, alpha, gamma, eps, lam, rows, rshape = ctx.consts
        gout = _f32c(g).reshape(1)
        need = ctx.needs_input_grad
        dxs = [torch.empty_like(x) if need[4 + i] else None for i, x in enumerate(xs)]
        dreg = torch.empty_like(rp) if need[3] else None
        Cs = (C.c_int * len(xs))(*[x.shape[1] for x in xs])
        pa = lambda ts_: (C.c_void_p * len(ts_))(*[ptr(t) for t in ts_])   # noqa: E731
        call("timhip_det_side_loss_bwd", pa(xs), pa(ts), Cs, len(xs), rows, ptr(u), ptr(off), ptr(rp), thr, alpha, gamma, eps, lam,
             ptr(block), ptr(gout), pa(dxs), ptr(dreg), _stream())
        return (None, None, None, dreg.view(rshape) if dreg is not None else None) + tuple(dxs)


def detection_side_loss(cls_logits, cls_targets, reg_pred, offsets, iou, normaliser, iou_threshold, lambda_reg=0.5, momentum=0.9,
                        alpha=0.25, gamma=2.0, eps=1e-8):
    """One modality side of the detection training loss, det scripts/train.py:222-349, as a handful of launches:

        valid_cls = iou >= 0;  w = where(iou < iou_threshold, 1, iou);  positive = offsets[:, 0] != inf
        normaliser <- momentum * normaliser + (1 - momentum) * max(#positive, 1)          (`normaliser`: device scalar, updated
                                                                                            IN PLACE - a replayed HIP graph advances it)
        loss = sum_k focal_sum(cls_logits[k], cls_targets[k], w, valid_cls) / (K * normaliser)
               + lambda_reg * diou_sum(reg_pred[positive], offsets[positive]) / normaliser      (when there is a positive row)

    cls_logits / cls_targets: lists of the side's K heads ([rows, C_k] each; verb / noun / action, or the audio head).  The loop's
    eager form derives the flags and weights with ~25 small torch launches per side; here the kernels derive them in place
    (timhip_det_side_loss_fwd / _bwd).  Gradients flow to the logits and to reg_pred."""
    if not isinstance(cls_logits, (list, tuple)):
        cls_logits, cls_targets = [cls_logits], [cls_targets]
    if normaliser.dtype != torch.float32 or normaliser.numel() != 1 or not normaliser.is_cuda:
        raise ValueError("normaliser: a float32 scalar tensor on the GPU (it is advanced in place)")
    aux = (list(cls_targets), normaliser, iou_threshold, alpha, gamma, eps, lambda_reg, momentum)
    return _DetSideLossFn.apply(aux, iou, offsets, reg_pred, *cls_logits)


def get_loss(criterion, pred, y, weights=None, reduction="mean"):
    """detection models/helpers/losses/loss.py:5-14, same arguments.  The focal criterion with row weights runs as one
    fused kernel; any other criterion takes the reference's generic route."""
    if criterion is sigmoid_focal_loss:
        if reduction == "sum":
            return focal_loss_sum(pred, y, weights, None)
        if reduction == "mean":
            return focal_loss_sum(pred, y, weights, None) / max(pred.numel(), 1)
        return _focal_elementwise(pred, y, weights, 0.25, 2.0)
    if criterion is ctr_diou_loss_1d and weights is None and reduction in ("sum", "mean"):
        return ctr_diou_loss_1d(pred, y, reduction=reduction)
    loss = criterion(pred, y)
    if weights is not None:
        loss = loss * weights[:, None]
    if reduction == "mean":
        return loss.mean()
    elif reduction == "sum":
        return loss.sum()
    return loss
